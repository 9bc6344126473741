mkdir -p gpurun_out/r03s
timeout 900 python -m pytest tests -m gpu -q -x -k "general or codecap or randomised or reference_defaults or every_window" 2>&1 | tail -3
timeout 500 python tools/stress_parity.py 500 41 2>&1 | tail -1
for a in "--window 5 3" "--window 3 1" "--code bb72 --window 3 1" "--code bb72" "--code hgp225 --window 3 1 --shots 16384" "--window 5 3 --osd-method osd_cs --osd-order 1"; do
  for nl in 1 0; do
    if [ $nl = 1 ]; then export QD_NO_LDS_EDGE=1; else unset QD_NO_LDS_EDGE; fi
    timeout 300 python bench.py --bp-method product_sum --schedule parallel --max-iter 10 $a --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('hbm_kernel=$nl $a', round(d['value']), round(d['ms_per_step'],1), d.get('logical_error_rate'), round(r['avg_launch_ms'],1), round(r['osd_kernel_ms_per_launch'],1))
"
  done
done | tee gpurun_out/r03s/ps_lds_ab.txt
