#!/bin/bash
# Build timing-ablation variants of the BP kernel (here, CPU) or run them (GPU box).  usage: ablate_bp.sh build|run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build_ablate
  for v in "0 8" "0 4" "1 8" "2 8" "3 8" "4 8" "12 8"; do set -- $v
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DQD_ABLATE=$1 -DQD_BP_MINWAVES=$2 \
      -o build_ablate/lib_a$1_w$2.so quits_amd/csrc/qd_api.hip quits_amd/csrc/bp_kernels.hip quits_amd/csrc/osd_kernels.hip quits_amd/csrc/gf2_kernels.hip &
  done; wait; ls build_ablate
else
  for f in build_ablate/*.so; do
    QUITS_AMD_LIB=$PWD/$f python bench.py --steps 2 --warmup 1 --no-cpu --shots 32768 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; it=d['mean_bp_iters']
print('$f', 'bp_ms %.2f' % r['avg_launch_ms'], 'iters %.2f' % it, 'us/shot-iter/GPU %.4f' % (1e3*r['avg_launch_ms']/(32768*it)))"
  done
fi
