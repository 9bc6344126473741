#!/bin/bash
# HIP priority of the post-processing stream (QD_POST_STREAM_PRIORITY, 0 = default, -1 = high), same box: usage tools/r06_postprio_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
B="--steps 3 --warmup 1 --no-cpu --no-api --no-other-configs"
for rep in 1 2; do
for pr in 0 -1; do
  export QD_POST_STREAM_PRIORITY=$pr
  for w in "headline|" "w3f1|--window 3 1 --shots 262144" "w5f3|--window 5 3 --shots 262144" "bb72w3f1|--code bb72 --window 3 1 --shots 1048576" "p6e-3|--p-override 0.006 --shots 262144" "osdcs1|--osd-method osd_cs --osd-order 1 --shots 131072" "lsdcs1|--osd-method lsd_cs --osd-order 1 --shots 262144"; do
    n=${w%%|*}; a=${w#*|}
    timeout 300 python bench.py $a $B 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('post stream priority $pr  %-9s' % '$n', 'shots/s', round(d['value']), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2), 'LER', round(d['logical_error_rate'],6))
" | tee -a $O/bench.txt
  done
done
done
