#!/bin/bash
# qd_decoder_post_head_start: the default (50 us, conditional) against off (QD_POST_HEAD_START_US=0), same box: usage tools/r06_headstart_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or public_call or p_sweep or lsd" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
B="--steps 2 --warmup 1 --no-cpu --no-api --no-other-configs"
for rep in 1 2 3; do
for us in 0 d; do
  if [ $us = d ]; then unset QD_POST_HEAD_START_US; else export QD_POST_HEAD_START_US=$us; fi
  for w in "headline|--steps 3" "p5e-3|--p 0.005 --shots 262144" "p4e-3|--p 0.004 --shots 524288"; do
    n=${w%%|*}; a=${w#*|}
    timeout 300 python bench.py $B $a 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('head start %-3s %-9s' % ('$us', '$n'), 'shots/s', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2), 'LER', round(d['logical_error_rate'],5))
" | tee -a $O/bench.txt
  done
done
done
