#!/bin/bash
# what the dependency chain of the serial schedule waits for: the default build against one whose suffix values do not come from memory (wrong results, timing only)
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
export QD_GEN_STAGES=0 SHOTS=16384,49152,81920 MAX_ITER=10,3
for v in default v_gen_nox; do
  L=""; [ $v != default ] && L=$PWD/build_ablate/lib_$v.so
  echo "== $v" | tee -a $O/curve.txt
  env ${L:+QUITS_AMD_LIB=$L} timeout 300 python tools/k1g_load_curve.py 2>&1 | grep "max_iter" | tee -a $O/curve.txt
done
