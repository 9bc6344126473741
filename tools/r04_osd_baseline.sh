#!/bin/bash
# round-4 baseline of the OSD kernels on one box: phase timers (QD_OSD_TIMING build) + bench lines
cd "$(dirname "$0")/.."
O=gpurun_out/r04a; mkdir -p $O
for f in bb144_custom_r12_p0.003 bb144_custom_r12_p0.006; do
  echo "== $f osd_0" >> $O/osd_timing.txt
  QUITS_AMD_LIB=$PWD/build_ablate/lib_osdtiming.so FIXTURE=$f timeout 300 python tools/osd_timing.py >> $O/osd_timing.txt 2>&1
done
echo "== headline osd_cs 1" >> $O/osd_timing.txt
QUITS_AMD_LIB=$PWD/build_ablate/lib_osdtiming.so OSD_METHOD=osd_cs OSD_ORDER=1 SHOTS=16384 timeout 300 python tools/osd_timing.py >> $O/osd_timing.txt 2>&1
for a in "--p 0.003" "--p 0.006" "--p 0.003 --osd-method osd_cs --osd-order 1 --shots 65536"; do
  QD_NO_PIPELINE=1 timeout 300 python bench.py $a --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$a', round(d['value']), round(d['ms_per_step'],1), d.get('logical_error_rate'), round(r['avg_launch_ms'],2), round(r['osd_kernel_ms_per_launch'],2), d.get('osd_frac'))
" >> $O/bench.txt
done
cat $O/osd_timing.txt $O/bench.txt
