#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in libquits_amd.so (hipcc -Rpass-analysis=kernel-resource-usage).

  python tools/resources.py [filter-substring ...] > profiles/rNN_kernel_resources.txt
Cross-compiles for gfx950; needs no GPU."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "quits_amd", "csrc")


def main():
    mk = open(os.path.join(CS, "Makefile")).read()
    src = re.search(r"^SRC := (.*)$", mk, re.M).group(1).split()
    flags = re.search(r"^FLAGS := (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    cmd = ["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/qd_res.so"] + src
    out = subprocess.run(cmd, cwd=CS, capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in out.splitlines():
        m = re.search(r"remark: (?:Function )?Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    dem = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    print("%-110s %5s %5s %5s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ"))
    for r, d in zip(rows, dem):
        d = re.sub(r"\(.*$", "", d).replace("HIP_vector_type<unsigned int, ", "uint<").replace("void ", "")
        if sys.argv[1:] and not any(f in d for f in sys.argv[1:]):
            continue
        print("%-110s %5d %5d %5d %7d %4d" % (d[:110], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("TotalSGPRs", r.get("SGPRs", -1)),
                                               r.get("ScratchSize", -1), r.get("Occupancy", r.get("Occupancy waves", -1))))


if __name__ == "__main__":
    main()
