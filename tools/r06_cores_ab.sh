#!/bin/bash
# does the BP kernel hide behind OSD-CS when both fit a CU?  pipelined driver on / off on windows of different LDS footprints: usage tools/r06_cores_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
B="--steps 2 --warmup 1 --no-cpu --no-api --no-other-configs --osd-method osd_cs --osd-order 1"
for w in "headline|--shots 262144" "w8f4|--window 8 4 --shots 262144" "w6f3|--window 6 3 --shots 262144" "w5f3|--window 5 3 --shots 262144"; do
  n=${w%%|*}; a=${w#*|}
  for np in 0 1; do
    if [ $np = 1 ]; then export QD_NO_PIPELINE=1; else unset QD_NO_PIPELINE; fi
    timeout 300 python bench.py $a $B 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-9s no_pipeline $np' % '$n', 'shots/s', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2), 'windows', len(d.get('config',{}).get('windows',[])) or '')
" | tee -a $O/bench.txt
  done
done
