#!/bin/bash
# a hold of N microseconds at the head of every BP stage (the post-processor of the other lane gets onto the CUs first): usage tools/r06_hold_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
B="--steps 2 --warmup 1 --no-cpu --no-api --no-other-configs"
for us in 0 30 100 400 2000; do
  if [ $us = 0 ]; then unset QD_BP_HEAD_DELAY_US; else export QD_BP_HEAD_DELAY_US=$us; fi
  for w in "headline|" "osdcs1|--osd-method osd_cs --osd-order 1 --shots 262144" "lsdcs1|--osd-method lsd_cs --osd-order 1 --shots 262144" "p6e-3|--p 0.006 --shots 262144" "w5f3|--window 5 3 --shots 393216"; do
    n=${w%%|*}; a=${w#*|}
    timeout 300 python bench.py $a $B 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('hold %5d us  %-9s' % ($us, '$n'), 'shots/s', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2), 'LER', round(d['logical_error_rate'],5))
" | tee -a $O/bench.txt
  done
done
