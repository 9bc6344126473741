#!/bin/bash
# the drop-in call's host path: pieces chained through one two-lane pipeline (default) against QD_NO_HOST_CHAIN=1, same box
# usage tools/r06_chain_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "public_call or streamed or host" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2; do
for v in "0 131072" "1 131072" "0 262144"; do
  set -- $v
  QD_NO_HOST_CHAIN=$1 QD_HOST_PIECE_SHOTS=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['through_api']; print('no_chain $1 piece $2', 'device-resident', round(d['value']), 'through_api warm', round(t['warm_shots_per_s']), 'cold_s', round(t['cold_s'],3), 'ratio', round(t['warm_over_device_resident'],3))
" | tee -a $O/bench.txt
done
done
