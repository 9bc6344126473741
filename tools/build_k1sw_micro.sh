#!/bin/bash
# A/B builds of instruction-level variants of the flooding min-sum kernel K1sw (bp_scatter_wide.hip); the other objects come from build/obj.
cd "$(dirname "$0")/.."
make -C quits_amd/csrc -s -j8 || exit 1
mkdir -p build_ablate
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed"
OTHER=$(ls build/obj/*.o | grep -v "bp_scatter_wide.o")
build() {
    local name=$1; shift
    local od=build_ablate/obj_k1sw_$name; mkdir -p $od
    /opt/rocm/bin/hipcc $CF "$@" -c -o $od/bp_scatter_wide.o quits_amd/csrc/bp_scatter_wide.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_ablate/lib_k1sw_$name.so $od/bp_scatter_wide.o $OTHER
}
build force -DQS_ABL_FORCE_ITERS &
build force_addrctl -DQS_ABL_FORCE_ITERS -DQS_ABL_ADDRCTL &
build force_noconf -DQS_ABL_FORCE_ITERS -DQS_ABL_NOCONF &
build scatpos -DQSW_SCAT_SGPR_POS=1 &
build rot -DQSW_GATHER_ROT=1 &
build rot_scatpos -DQSW_GATHER_ROT=1 -DQSW_SCAT_SGPR_POS=1 &
wait
ls -la build_ablate | grep "lib_k1sw_"
