# A/B of two builds of the library through bench.py (product-sum rows): QUITS_AMD_LIB=<old> vs the in-tree library
OLD=${1:-build_ablate/libold.so}
for a in "--schedule serial --window 5 3 --osd-method osd_cs --osd-order 1" "--schedule serial --window 3 1 --osd-method osd_cs --osd-order 1" "--schedule serial" "--schedule parallel --window 5 3" "--schedule parallel --window 3 1" "--schedule parallel --code bb72"; do
  for lib in "$OLD" ""; do
    if [ -n "$lib" ]; then export QUITS_AMD_LIB=$PWD/$lib; else unset QUITS_AMD_LIB; fi
    timeout 300 python bench.py --bp-method product_sum --max-iter 10 $a --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('lib=${lib:-tree} $a', round(d['value']), round(d['ms_per_step'],1), d.get('logical_error_rate'), d.get('mean_bp_iters'), round(r['avg_launch_ms'],1), round(r['osd_kernel_ms_per_launch'],1))
"
  done
done
