#!/bin/bash
# Instrumented builds for tools/osdcs_timing.py: build_ablate/lib_cstiming{1,2,3}.so = -DQD_OSD_TIMING -DQD_CS_SUB={1: panel phase, 2: sort, 3: sweep}
cd "$(dirname "$0")/.."
mkdir -p build_ablate
SRC=$(ls quits_amd/csrc/*.hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-pass-failed -Iinclude"
for m in ${1:-1 2 3}; do /opt/rocm/bin/hipcc $FLAGS -DQD_OSD_TIMING -DQD_CS_SUB=$m -o build_ablate/lib_cstiming$m.so $SRC & done
wait; ls -la build_ablate
