#!/bin/bash
# BP stage alone (65 536 headline shots) for instruction-level variants of K1sw, same box: tools/r05_k1sw_micro_ab.sh <tag> <variants...>
set -u
TAG=${1:-r05m}; shift
VARS=${*:-"before_hp_pairs default scatpos rot rot_scatpos"}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do
  for v in $VARS; do
    L=""; [ $v != default ] && L=$PWD/build_ablate/lib_k1sw_$v.so
    env ${L:+QUITS_AMD_LIB=$L} K1_STAGES=1 python tools/k1_time.py 2>&1 | grep "stage 1" | sed "s#^.*stage 1:#$v:#"
  done
done > $O/k1sw_micro_ab.txt 2>&1
cat $O/k1sw_micro_ab.txt
