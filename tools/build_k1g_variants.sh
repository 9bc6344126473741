#!/bin/bash
# A/B builds of the per-edge BP kernel's serial schedule (bp_general.hip): a column-weight-6 instantiation (QD_GEN_D6) and a register budget of
# six wavefronts per SIMD (QD_GEN_WPE=6).  Only bp_general.hip and qd_api.hip depend on the switches; the other objects come from build/obj.
cd "$(dirname "$0")/.."
make -C quits_amd/csrc -s -j8 || exit 1
mkdir -p build_ablate
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed"
OTHER=$(ls build/obj/*.o | grep -v "bp_general.o\|qd_api.o")
build() {   # name flags...
    local name=$1; shift
    local od=build_ablate/obj_$name; mkdir -p $od
    /opt/rocm/bin/hipcc $CF "$@" -c -o $od/bp_general.o quits_amd/csrc/bp_general.hip &&
    /opt/rocm/bin/hipcc $CF "$@" -c -o $od/qd_api.o quits_amd/csrc/qd_api.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_ablate/lib_k1g_$name.so $od/bp_general.o $od/qd_api.o $OTHER
}
build d6 -DQD_GEN_D6=1 &
build d6w6 -DQD_GEN_D6=1 -DQD_GEN_WPE=6 &
build d8w6 -DQD_GEN_WPE=6 &
wait
ls -la build_ablate | grep k1g
