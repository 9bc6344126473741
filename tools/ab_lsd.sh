#!/bin/bash
# same-box A/B of two library builds on the BP-LSD headline run (GPU box).  usage: tools/ab_lsd.sh <libA.so> <libB.so>
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then lib=$1; else lib=$2; fi
    env QUITS_AMD_LIB=$PWD/$lib timeout 300 python bench.py --osd-method lsd_0 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['osd_kernel_ms_per_launch'],2), d['logical_error_rate'])"
  done
done
