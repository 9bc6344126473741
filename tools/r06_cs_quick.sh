#!/bin/bash
# Quick check of the OSD-CS / OSD-E kernel (osd_cs.hip) on one GPU box: parity tests that reach it, a short randomised sweep, phase timers, three bench points.
# usage (through gpurun): bash tools/r06_cs_quick.sh <tag> [stress trials]
TAG=${1:-r06cs}
O=gpurun_out/$TAG
mkdir -p $O
make -C oracle -s 2>&1 | tail -2
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "higher_order_osd or other_postprocessors or config4_qlp_sliding_window_bit_exact or osdw_panel or osd_alone or productsum_serial_osdcs" > $O/tests.txt 2>&1
tail -4 $O/tests.txt
timeout 400 python tools/stress_parity.py ${2:-300} 606 > $O/stress.txt 2>&1
tail -3 $O/stress.txt
B="--steps 3 --warmup 1 --no-cpu --no-api --no-other-configs"
timeout 300 python bench.py --osd-method osd_cs --osd-order 1 --shots 131072 $B > $O/headline_osdcs1.json 2>> $O/err.txt
timeout 300 python bench.py --code qlp1020 --window 3 1 --p-override 0.001 --osd-method osd_cs --osd-order 1 --shots 8192 --steps 2 --warmup 1 --no-cpu --no-api --no-other-configs > $O/qlp_w3f1_osdcs1.json 2>> $O/err.txt
timeout 300 python bench.py --window 5 3 --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --shots 163840 $B > $O/refsettings_w5f3.json 2>> $O/err.txt
timeout 300 python bench.py --osd-method osd_e --osd-order 8 --shots 131072 $B > $O/headline_osde8.json 2>> $O/err.txt
python - <<PY | tee $O/summary.txt
import json, glob
print("%-28s %12s %10s %10s %10s  %s" % ("workload", "shots/s", "LER", "BP ms", "post ms", "post kernel"))
for f in sorted(glob.glob("$O/*.json")):
    try:
        o = json.load(open(f)); r = o["roofline"]
        print("%-28s %12.0f %10.6f %10.2f %10.2f  %s" % (f.split("/")[-1][:-5], o["value"], o["logical_error_rate"], r.get("avg_launch_ms") or 0, r.get("osd_kernel_ms_per_launch") or 0, r.get("osd", {}).get("kernel")))
    except Exception as e:
        print(f, "ERR", e)
PY
if [ -f build_ablate/lib_cstiming1.so ]; then
  QD_CS_SUB=1 QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming1.so timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase_headline_sub1.txt
  QD_CS_SUB=1 QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming1.so FIXTURE=qlp1020_cardinal_r20_p0.003 WINDOW=3,1,5 SHOTS=2048 timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase_qlp_sub1.txt
  cat $O/phase_headline_sub1.txt
fi
for sub in 4 5; do
if [ -f build_ablate/lib_cstiming$sub.so ]; then
  QD_CS_SUB=$sub QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming$sub.so timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase_headline_sub$sub.txt
  tail -4 $O/phase_headline_sub$sub.txt
fi
done
