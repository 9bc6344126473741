#!/bin/bash
# headline bench with alternative builds of the library, same box: usage tools/r06_lib_ab.sh <outdir> <lib1> <lib2> ...   ("default" = the in-tree library)
cd "$(dirname "$0")/.."
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2 3; do
for v in "$@"; do
  L=""; [ $v != default ] && L=$PWD/build_ablate/lib_$v.so
  env ${L:+QUITS_AMD_LIB=$L} timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-api --no-other-configs 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'shots/s', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2), 'LER', d['logical_error_rate'])
" | tee -a $O/bench.txt
done
done
