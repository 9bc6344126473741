# same-box A/B of the serial BP kernel through bench.py: an environment switch (default QD_NO_LDS_PREFIX=1 = prefixes in the HBM plane) against the default path
SW=${1:-QD_NO_LDS_PREFIX}
timeout 600 python -m pytest tests -m gpu -q -x -k "general or serial or reference_defaults or randomised" 2>&1 | tail -2
for a in "--schedule serial --window 5 3 --osd-method osd_cs --osd-order 1" "--schedule serial --window 3 1" "--schedule serial" "--bp-method minimum_sum --schedule serial" "--schedule serial --window 5 3 --shots 81920" "--schedule serial --code bb72 --window 3 1"; do
  for sw in 1 0 1 0; do
    if [ $sw = 1 ]; then export $SW=1; else unset $SW; fi
    timeout 300 python bench.py --bp-method product_sum --max-iter 10 $a --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$SW=$sw $a', round(d['value']), round(d['ms_per_step'],1), d.get('logical_error_rate'), round(r['avg_launch_ms'],1), round(r['osd_kernel_ms_per_launch'],1))
"
  done
done
