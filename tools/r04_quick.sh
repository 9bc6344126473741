#!/bin/bash
# quick GPU check of the OSD-0 path: parity tests that touch it, then bench lines at p = 3e-3 / 6e-3 (one stream)
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r04q}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "${2:-osd or scatter or config3 or inconsistent or rank or window_shape or randomised}" 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
for a in "--p 0.003" "--p 0.006"; do
  for sw in 0 1; do
  QD_NO_OSD_SR=$sw QD_NO_PIPELINE=1 timeout 300 python bench.py $a --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('NO_SR=$sw $a', round(d['value']), round(d['ms_per_step'],1), d.get('logical_error_rate'), round(r['avg_launch_ms'],2), round(r['osd_kernel_ms_per_launch'],2), d.get('osd_frac'))
" | tee -a $O/bench.txt
  done
done
