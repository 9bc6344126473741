#!/bin/bash
# Wrong-result timing builds of the flooding min-sum kernel K1sw (bp_scatter_wide.hip) for tools/ab_variants.sh: every shot is forced through max_iter
# iterations (QS_ABL_FORCE_ITERS), and the scatter pass's LDS atomic is replaced by a plain store / a plain read / nothing.
cd "$(dirname "$0")/.."
mkdir -p build_ablate
SRC=$(ls quits_amd/csrc/*.hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-pass-failed -Iinclude"
/opt/rocm/bin/hipcc $FLAGS -DQS_ABL_FORCE_ITERS -o build_ablate/lib_k1sw_force.so $SRC &
/opt/rocm/bin/hipcc $FLAGS -DQS_ABL_FORCE_ITERS -DQS_ABL_STORE -o build_ablate/lib_k1sw_force_store.so $SRC &
/opt/rocm/bin/hipcc $FLAGS -DQS_ABL_FORCE_ITERS -DQS_ABL_READ -o build_ablate/lib_k1sw_force_read.so $SRC &
/opt/rocm/bin/hipcc $FLAGS -DQS_ABL_FORCE_ITERS -DQS_ABL_NOADD -o build_ablate/lib_k1sw_force_noadd.so $SRC &
wait; ls -la build_ablate | grep k1sw
