#!/bin/bash
# variant builds of the whole library with other switches of qd_osdcs_kernel: build_ablate/lib_v_<name>.so
# usage: tools/build_cs_variants.sh "name:-DQD_CS_PUSH_MERGED=0" "prio1:-DQD_CS_B_PRIO=1" ...
cd "$(dirname "$0")/../quits_amd/csrc"
mkdir -p ../../build_ablate
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( make -s -j3 OBJDIR=../../build/obj_v_$name OUT=../../build_ablate/lib_v_$name.so EXTRA="$flags" 2>&1 | grep -E "error" ) &
done
wait
ls -la ../../build_ablate | grep lib_v_
