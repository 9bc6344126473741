#!/bin/bash
# variant builds of the whole library with different shapes of qd_osd0_sr_kernel: build_ablate/lib_sr_<name>.so
# usage: tools/build_sr_variants.sh "name:make-args" ...   e.g. "w4:EXTRA=-DQD_SR_WPS=4" "w4ns:EXTRA=-DQD_SR_WPS=4 SAFE_SPILL="
cd "$(dirname "$0")/../quits_amd/csrc"
mkdir -p ../../build_ablate
for spec in "$@"; do
  name=${spec%%:*}; margs=${spec#*:}
  ( eval make -s -j3 OBJDIR=../../build/obj_$name OUT=../../build_ablate/lib_sr_$name.so $margs 2>&1 | grep -E "error" ) &
done
wait
ls -la ../../build_ablate | grep lib_sr
