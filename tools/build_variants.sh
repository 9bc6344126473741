#!/bin/bash
# Instrumented builds of libquits_amd.so for the phase-timer scripts (run here; build_ablate/ travels to the GPU box).
#   lib_bptiming.so  -DQD_BP_TIMING   -> tools/bp_timing.py      lib_osdtiming.so -DQD_OSD_TIMING -> tools/osd_timing.py
#   valu_rate                         -> tools/ubench/valu_rate.hip
cd "$(dirname "$0")/.."
mkdir -p build_ablate
SRC="quits_amd/csrc/qd_api.hip quits_amd/csrc/bp_kernels.hip quits_amd/csrc/bp_general.hip quits_amd/csrc/osd_kernels.hip quits_amd/csrc/lsd_kernels.hip quits_amd/csrc/gf2_kernels.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude"
/opt/rocm/bin/hipcc $FLAGS -DQD_BP_TIMING -o build_ablate/lib_bptiming.so $SRC &
/opt/rocm/bin/hipcc $FLAGS -DQD_OSD_TIMING -o build_ablate/lib_osdtiming.so $SRC &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o build_ablate/valu_rate tools/ubench/valu_rate.hip &
wait; ls -la build_ablate
