#!/bin/bash
# chunk size of the pipelined driver on the headline (QD_CHUNK_SHOTS), same box: usage tools/r06_chunk_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
for c in 65536 131072 262144 32768; do
  QD_CHUNK_SHOTS=$c timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-api --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('chunk $c', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'bp', round(r['avg_launch_ms'],2), 'osd', round(r['osd_kernel_ms_per_launch'],2), d.get('logical_error_rate'))
" | tee -a $O/bench.txt
done
done
