#!/bin/bash
# shots per staged piece of the drop-in call's host path (QD_HOST_PIECE_SHOTS), same box: usage tools/r06_piece_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
for c in 262144 524288 131072 1048576; do
  QD_HOST_PIECE_SHOTS=$c timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['through_api']; print('piece $c', 'device-resident', round(d['value']), 'through_api warm', round(t['warm_shots_per_s']), 'cold_s', round(t['cold_s'],3), 'ratio', round(t['warm_over_device_resident'],3))
" | tee -a $O/bench.txt
done
done
