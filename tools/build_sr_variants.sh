#!/bin/bash
# variant builds of the whole library with different shapes of qd_osd0_sr_kernel: build_ablate/lib_sr_<name>.so (+ _t = with phase timers)
cd "$(dirname "$0")/../quits_amd/csrc"
mkdir -p ../../build_ablate
build() { name=$1; shift; make -s -j2 OBJDIR=../../build/obj_$name OUT=../../build_ablate/lib_sr_$name.so EXTRA="$*" 2>&1 | grep -E "error|warning: v" ; }
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build $name $flags &
  build ${name}_t $flags -DQD_OSD_TIMING &
  wait
done
ls -la ../../build_ablate | grep lib_sr
