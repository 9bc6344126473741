#!/bin/bash
# per-launch durations of the staged serial schedule (rocprofv3 kernel trace): usage tools/r06_stage_trace.sh <outdir> <bench args...>
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py "$@" --steps 1 --warmup 1 --no-cpu --no-api --no-other-configs > $O/bench.json 2> $O/err.txt
python - <<PY > $O/launches.txt
import csv, glob
f = sorted(glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-60:]:
    n = r["Kernel_Name"][:60]
    print("%10.3f ms  +%9.3f ms  grid %8s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?")), n))
PY
tail -45 $O/launches.txt
