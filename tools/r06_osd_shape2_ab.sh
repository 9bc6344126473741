#!/bin/bash
# OSD-0 workgroups per CU and chunk size under the round's final driver (no fills between BP kernels), headline: usage tools/r06_osd_shape2_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
B="--steps 3 --warmup 1 --no-cpu --no-api --no-other-configs"
for rep in 1 2; do
for v in "default|" "sr_per_cu_1|QD_OSD_SR_PER_CU=1" "sr_per_cu_2|QD_OSD_SR_PER_CU=2" "sr_per_cu_3|QD_OSD_SR_PER_CU=3" "chunk_131072|QD_CHUNK_SHOTS=131072" "chunk_32768|QD_CHUNK_SHOTS=32768"; do
  n=${v%%|*}; e=${v#*|}
  env $e timeout 300 python bench.py $B 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s' % '$n', 'shots/s', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2), 'LER', round(d['logical_error_rate'],5))
" | tee -a $O/bench.txt
done
done
