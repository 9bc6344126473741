#!/bin/bash
# PMC passes over bench.py focused on the BP/OSD kernels (run through gpurun).  usage: tools/pmc_bp.sh <tag>
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu --shots 32768"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d $OUT/p$i -o p$i -- python $REPO/bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || tail -3 $OUT/p$i.err
done
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
        if not k.startswith("qd_"): continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES","SQ_LDS_BANK_CONFLICT","SQ_ACTIVE_INST_VMEM","GRBM_GUI_ACTIVE"): cnt[k+"/"+r["Counter_Name"]] += 1
lines=[]
for k, d in per.items():
    lines.append("== " + k)
    for c, v in sorted(d.items()): lines.append("   %-28s %18.0f" % (c, v))
txt="\n".join(lines); print(txt); open("$OUT/summary.txt","w").write(txt+"\n")
PY
