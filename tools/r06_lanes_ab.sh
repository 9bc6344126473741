#!/bin/bash
# lanes of the pipelined driver (QD_PIPELINE_LANES), same box: usage tools/r06_lanes_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or public_call or sliding or window" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
B="--steps 3 --warmup 1 --no-cpu --no-other-configs"
for rep in 1 2; do
for ln in 2 3 4; do
  export QD_PIPELINE_LANES=$ln
  for w in "headline|" "w3f1|--window 3 1 --shots 393216" "w5f3|--window 5 3 --shots 393216" "bb72w3f1|--code bb72 --window 3 1 --shots 1572864" "qlp_w3f1|--code qlp1020 --window 3 1 --p-override 0.001 --shots 24576 --no-api"; do
    n=${w%%|*}; a=${w#*|}
    timeout 300 python bench.py $a $B 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=d.get('through_api') or {}; print('lanes $ln  %-9s' % '$n', 'shots/s', round(d['value']), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2), 'LER', round(d['logical_error_rate'],6), 'through_api', round(t.get('warm_shots_per_s') or 0))
" | tee -a $O/bench.txt
  done
done
done
