#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile_bench.sh into one text file (copied under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("# kernel stats (%s)" % os.path.relpath(f, out))
    lines += [ln.rstrip() for ln in open(f)][:12]
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    per = defaultdict(lambda: [0.0, 0, 0.0])
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                k = row["Kernel_Name"].split("(")[0][:60]
                per[k][0] += float(row["Counter_Value"])
                per[k][1] += 1
                per[k][2] = max(per[k][2], float(row["Counter_Value"]))
    lines.append("# %s per kernel: total (counter units: KiB), dispatches, mean per dispatch, largest dispatch" % counter)
    for k, (v, c, mx) in sorted(per.items(), key=lambda kv: -kv[1][0])[:8]:
        lines.append("%-60s %16.1f %6d %16.1f %16.1f" % (k, v, c, v / max(c, 1), mx))
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
