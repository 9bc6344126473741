#!/usr/bin/env python3
"""Timeline of the last N kernel launches of a rocprofv3 kernel trace (csv): start, duration, gap to the previous launch's end on the same queue.
usage: tools/r06_timeline.py <trace dir> [N]"""
import csv, glob, sys
d = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
out = []
for r in rows:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    out.append("%11.3f ms  +%9.3f ms  gap %9.1f us  q %-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, q, r["Kernel_Name"][:48]))
print("\n".join(out[-N:]))
