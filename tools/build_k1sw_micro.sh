#!/bin/bash
# A/B / ablation builds of the flooding min-sum kernel K1sw (bp_scatter_wide.hip); the other objects come from build/obj.
#   tools/build_k1sw_micro.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/.."
make -C quits_amd/csrc -s -j8 || exit 1
mkdir -p build_ablate
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed"
OTHER=$(ls build/obj/*.o | grep -v "bp_scatter_wide.o")
build() {
    local name=$1; shift
    local od=build_ablate/obj_k1sw_$name; mkdir -p $od
    /opt/rocm/bin/hipcc $CF $@ -c -o $od/bp_scatter_wide.o quits_amd/csrc/bp_scatter_wide.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_ablate/lib_k1sw_$name.so $od/bp_scatter_wide.o $OTHER
}
while [ $# -ge 2 ]; do build "$1" $2 & shift 2; done
wait
ls -la build_ablate | grep "lib_k1sw_"
