#!/bin/bash
# headline through the PIPELINED driver with other builds of the library (QUITS_AMD_LIB): usage tools/r06_pipe_ab.sh <outdir> <libname> [<libname> ...]  ("main" = quits_amd/lib; else build_ablate/lib_<name>.so)
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O; shift
for rep in 1 2; do
for v in "$@"; do
  if [ $v = main ]; then unset QUITS_AMD_LIB; else export QUITS_AMD_LIB=$PWD/build_ablate/lib_$v.so; fi
  for p in 0.003 0.005; do
  timeout 300 python bench.py --p $p --steps 6 --warmup 2 --no-cpu --no-api --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v p=$p', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'bp', round(r['avg_launch_ms'],2), 'osd', round(r['osd_kernel_ms_per_launch'],2), d.get('logical_error_rate'))
" | tee -a $O/bench.txt
  done
done
done
