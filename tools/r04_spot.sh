#!/bin/bash
# spot configs on one box: tools/r04_spot.sh <tag>
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r04spot}; mkdir -p $O
for a in "--osd-method lsd_cs --osd-order 1" "--osd-method osd_cs --osd-order 1 --steps 2" "--window 5 3" "--window 3 1" "--code bb72" "--code bb72 --window 3 1" "--code qlp1020 --window 3 1 --shots 8192 --p-override 0.001 --steps 2" "--code qlp1020 --window 3 1 --shots 8192 --p-override 0.001 --steps 2 --osd-method osd_cs --osd-order 1" "--p 0.006" "--bp-method product_sum --schedule serial --max-iter 10 --window 5 3 --osd-method osd_cs --osd-order 1 --shots 81920 --steps 2"; do
  timeout 600 python bench.py --no-cpu --no-api --steps 3 --warmup 1 $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$a |', round(d['value']), 'shots/s | ms/step', round(d['ms_per_step'],1), '| pL', round(d.get('logical_error_rate'),5), '| bp ms', round(r['avg_launch_ms'],2), '| osd ms', round(r['osd_kernel_ms_per_launch'],2))"
done | tee $O/spot_configs.txt
