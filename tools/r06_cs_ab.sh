#!/bin/bash
# same-box comparison of qd_osdcs_kernel builds on the headline window with OSD-CS(1): usage tools/r06_cs_ab.sh <outdir> <name> [<name> ...]  ("main" = quits_amd/lib)
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O; shift
for rep in 1 2; do
for v in "$@"; do
  if [ $v = main ]; then unset QUITS_AMD_LIB; else export QUITS_AMD_LIB=$PWD/build_ablate/lib_cs_$v.so; fi
  QD_NO_PIPELINE=1 timeout 300 python bench.py --osd-method osd_cs --osd-order 1 --shots 131072 --steps 3 --warmup 1 --no-cpu --no-api --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value']), 'bp', round(r['avg_launch_ms'],2), 'osdcs', round(r['osd_kernel_ms_per_launch'],2), d.get('logical_error_rate'))
" | tee -a $O/bench.txt
done
done
