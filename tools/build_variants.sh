#!/bin/bash
# Instrumented builds of libquits_amd.so for the phase-timer scripts (run here; build_ablate/ travels to the GPU box).
#   lib_bptiming.so  -DQD_BP_TIMING   -> tools/bp_timing.py      lib_osdtiming.so -DQD_OSD_TIMING -> tools/osd_timing.py
#   usage: tools/build_variants.sh [osd|bp|all]
cd "$(dirname "$0")/.."
mkdir -p build_ablate
SRC=$(ls quits_amd/csrc/*.hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-pass-failed -Iinclude"
what=${1:-all}
if [ $what = bp ] || [ $what = all ]; then /opt/rocm/bin/hipcc $FLAGS -DQD_BP_TIMING -o build_ablate/lib_bptiming.so $SRC & fi
if [ $what = osd ] || [ $what = all ]; then /opt/rocm/bin/hipcc $FLAGS -DQD_OSD_TIMING -o build_ablate/lib_osdtiming.so $SRC & fi
wait; ls -la build_ablate
