#!/bin/bash
# same-box A/B of two library builds on the headline and p = 6e-3 points (GPU box).  usage: tools/ab_osd.sh <libA.so> <libB.so> [env for B]
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then lib=$1; extra=""; else lib=$2; extra="$3"; fi
    for p in 0.003 0.006; do
      env QUITS_AMD_LIB=$PWD/$lib $extra timeout 200 python bench.py --p $p --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$p', round(d['value']), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['osd_kernel_ms_per_launch'],2))"
    done
  done
done
