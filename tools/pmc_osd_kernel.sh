#!/bin/bash
# SQ counters of single OSD launches (32768 headline shots: BP + OSD-0, the OSD kernels' rows are kept).
# usage (through gpurun): tools/pmc_osd_kernel.sh <tag> [FIXTURE] [OSD_METHOD OSD_ORDER]
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcosd_$TAG
mkdir -p $OUT
export FIXTURE=${2:-bb144_custom_r12_p0.003} OSD_METHOD=${3:-osd_0} OSD_ORDER=${4:-0}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_CYCLES SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d $OUT/p$i -o p$i -- python $REPO/tools/osd_timing.py > $OUT/p$i.log 2> $OUT/p$i.err || tail -3 $OUT/p$i.err
done
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ","")[:48]
        if not k.startswith("qd_osd"): continue
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines=["# $FIXTURE $OSD_METHOD $OSD_ORDER: tools/osd_timing.py (32768 shots, two decode calls); per launch: the largest launch | number of launches",
       "# (FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE counts half the traffic on gfx950: MI355X_MICROARCH.md)"]
for k, d in per.items():
    lines.append("== " + k)
    for c, v in sorted(d.items()): lines.append("   %-28s %18.0f   %d" % (c, max(v), len(v)))
txt="\n".join(lines); print(txt); open("$OUT/summary.txt","w").write(txt+"\n")
PY
