#!/bin/bash
# multi-window plans: pipelined driver (three lanes) against one stream, same box: usage tools/r06_pipe_multi_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
B="--steps 2 --warmup 1 --no-cpu --no-api --no-other-configs"
for rep in 1 2; do
for np in 0 1; do
  if [ $np = 1 ]; then export QD_NO_PIPELINE=1; else unset QD_NO_PIPELINE; fi
  for w in "w3f1|--window 3 1 --shots 393216" "w5f3|--window 5 3 --shots 393216" "bb72w3f1|--code bb72 --window 3 1 --shots 786432" "bb72|--code bb72 --shots 786432" "qlp|--code qlp1020 --window 3 1 --p-override 0.001 --shots 24576"; do
    n=${w%%|*}; a=${w#*|}
    timeout 300 python bench.py $B $a 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('one stream %d  %-9s' % ($np, '$n'), 'shots/s', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2))
" | tee -a $O/bench.txt
  done
done
done
