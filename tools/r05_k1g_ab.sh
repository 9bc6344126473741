#!/bin/bash
# A/B of the serial per-edge kernel's register budget / unroll bound / prefix storage on one GPU box: tools/r05_k1g_ab.sh <tag>
set -u
TAG=${1:-r05z}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
export MAX_ITER=10
for lib in default d6 d6w6 d8w6; do
  for nolp in 0 1; do
    L=""; [ $lib != default ] && L=$PWD/build_ablate/lib_k1g_$lib.so
    E=""; [ $nolp = 1 ] && E="QD_NO_LDS_PREFIX=1"
    echo "== lib $lib  no_lds_prefix $nolp"
    env ${L:+QUITS_AMD_LIB=$L} $E SHOTS=81920,98304,196608 python tools/k1g_load_curve.py 2>&1 | grep "max_iter"
  done
done > $O/k1g_ab.txt 2>&1
cat $O/k1g_ab.txt
