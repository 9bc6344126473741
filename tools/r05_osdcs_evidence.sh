#!/bin/bash
# Evidence set for the rebuilt OSD-CS / OSD-E kernel (osd_cs.hip): A/B against the kernel it replaces (QD_OSDCS_OLD=1), phase timers, SQ counters.
# usage (through gpurun): bash tools/r05_osdcs_evidence.sh <tag>     -> gpurun_out/<tag>/
TAG=${1:-r05c}
O=gpurun_out/$TAG
mkdir -p $O
B="--steps 3 --warmup 1 --no-cpu --no-api --no-other-configs"
for old in 0 1; do
  QD_OSDCS_OLD=$old python bench.py --osd-method osd_cs --osd-order 1 --shots 131072 $B > $O/ab_headline_osdcs1_old$old.json 2>> $O/err.txt
  QD_OSDCS_OLD=$old python bench.py --code qlp1020 --window 3 1 --p-override 0.001 --osd-method osd_cs --osd-order 1 --shots 8192 --steps 2 --warmup 1 --no-cpu --no-api --no-other-configs > $O/ab_qlp_w3f1_osdcs1_old$old.json 2>> $O/err.txt
  QD_OSDCS_OLD=$old python bench.py --window 5 3 --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --shots 163840 $B > $O/ab_refsettings_w5f3_old$old.json 2>> $O/err.txt
  QD_OSDCS_OLD=$old python bench.py --code hgp225 --window 3 1 --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --shots 65536 $B > $O/ab_hgp225_w3f1_old$old.json 2>> $O/err.txt
  QD_OSDCS_OLD=$old python bench.py --osd-method osd_e --osd-order 8 --shots 131072 $B > $O/ab_headline_osde8_old$old.json 2>> $O/err.txt
done
python - <<PY > $O/ab_summary.txt
import json, glob
print("# new = qd_osdcs_kernel (osd_cs.hip), old = QD_OSDCS_OLD=1 (qd_osdw_col_kernel / row form); same box, same shots")
print("%-34s %12s %10s %12s %12s  %s" % ("workload", "shots/s", "LER", "BP ms", "post ms", "post kernel"))
for f in sorted(glob.glob("$O/ab_*.json")):
    try:
        o = json.load(open(f)); r = o["roofline"]
        print("%-34s %12.0f %10.6f %12.2f %12.2f  %s" % (f.split("/")[-1][3:-5], o["value"], o["logical_error_rate"], r.get("avg_launch_ms") or 0, r.get("osd_kernel_ms_per_launch") or 0, r.get("osd", {}).get("kernel")))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/ab_summary.txt
for m in 1 2 3; do
  QD_CS_SUB=$m QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming$m.so timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase_headline_sub$m.txt
done
QD_CS_SUB=1 QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming1.so FIXTURE=qlp1020_cardinal_r20_p0.003 WINDOW=3,1,5 SHOTS=2048 timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase_qlp_sub1.txt
QD_CS_SUB=1 QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming1.so WINDOW=5,3,1 timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase_w5f3_sub1.txt
cat $O/phase_headline_sub1.txt
bash tools/pmc_osd_kernel.sh ${TAG}_cs1 bb144_custom_r12_p0.003 osd_cs 1 > $O/pmc_headline_osdcs1.txt 2>&1
tail -30 $O/pmc_headline_osdcs1.txt
