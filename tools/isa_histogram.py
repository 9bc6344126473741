#!/usr/bin/env python3
"""Instruction-class histogram of the flooding min-sum kernel's loops, from the compiler's own assembly (hipcc -S), so that the
issue-bound figures in bench.py / DESIGN.md can be reproduced from profiles/ (VERDICT r01, item 3).

  python tools/isa_histogram.py [--kernel SUBSTR] [--json out.json] > profiles/rNN_k1_isa_histogram.txt

Classes (issue cost per wavefront per SIMD measured by tools/ubench/valu_rate.hip, profiles/r01f_valu_issue_rates.txt):
  valu_fast  ~2 clk : VOP1/VOP2 add, sub, mul, and, or, xor, shifts, mov, not, fma/fmac, cvt
  valu_slow  ~4 clk : compares, v_cndmask, min/max/med3, three-operand logic (lshl_or, and_or, bfi, bfe, alignbit, lshl_add),
                      64-bit shifts / adds, SDWA / DPP forms, v_readfirstlane / v_readlane, mad / mul_lo / mul_hi
  salu, lds (ds_*), vmem (global_/buffer_/scratch_/flat_), branch, wait (s_waitcnt, s_nop, s_barrier)
Cross-compiles for gfx950; needs no GPU."""
import argparse, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "quits_amd", "csrc")
DEFAULT = "qd_bp_minsum_kernelILi1024ELi2ELi1E15HIP_vector_typeIjLj4EELi8E"      # the headline window's instantiation

SLOW = re.compile(r"^v_(cmp|cmpx|cndmask|min|max|med3|lshl_or|and_or|or3|xad|bitop3|bfi|bfe|alignbit|alignbyte|lshl_add|add_lshl|add3|"
                  r"lshlrev_b64|lshrrev_b64|ashrrev_i64|readfirstlane|readlane|writelane|mad_|mul_lo|mul_hi|perm|sad|pk_|"
                  r"add_co|sub_co|addc|subb|add_f64|fma_f64|mul_f64|rcp|rsq|sqrt|exp|log|sin|cos|ldexp|frexp|div_|trig)")


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if "sdwa" in ins or "dpp" in ins or "_sdwa" in op or "_dpp" in op or SLOW.match(op):
            return "valu_slow"
        return "valu_fast"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm") or op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.split("_")[0] in ("global", "buffer", "scratch", "flat"):
        return "vmem"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=DEFAULT)
    ap.add_argument("--src", default="bp_kernels.hip")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    mk = open(os.path.join(CS, "Makefile")).read()
    flags = re.search(r"^FLAGS := (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    flags = [f for f in flags if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", out, a.src], cwd=CS, check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and a.kernel in l and l.rstrip().endswith(
        tuple([":"])) or (l.startswith("_Z") and a.kernel in l and ": ;" in l))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section"))
    body = lines[start:end]
    # basic blocks with the loop depth LLVM annotates
    blocks, cur = [], {"label": "entry", "depth": 0, "ins": [], "inner": False}
    for l in body[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "depth": 0, "ins": [], "inner": False}
            d = re.search(r"Depth=(\d+)", l)
            if d:
                cur["depth"] = int(d.group(1)); cur["inner"] = "Inner Loop Header" in l
            elif "in Loop: Header" in l:
                cur["depth"] = -1           # inside a loop, depth given by the header comment of that loop
            continue
        t = l.strip()
        if t.startswith(";") and "Depth=" in t and not cur["ins"]:
            d = re.search(r"Depth=(\d+)", t)
            cur["depth"] = max(cur["depth"], int(d.group(1)))
            cur["inner"] = cur["inner"] or "Inner Loop Header" in t
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    blocks.append(cur)
    classes = ["valu_fast", "valu_slow", "salu", "lds", "vmem", "branch", "wait", "other"]
    print("# kernel %s" % a.kernel)
    print("# %-14s %5s %5s | %s" % ("block", "depth", "inner", " ".join("%9s" % c for c in classes)))
    rows = []
    for b in blocks:
        if not b["ins"]:
            continue
        cnt = {c: 0 for c in classes}
        for ins in b["ins"]:
            cnt[classify(ins)] += 1
        rows.append((b, cnt))
        print("  %-14s %5d %5s | %s" % (b["label"], b["depth"], "yes" if b["inner"] else "", " ".join("%9d" % cnt[c] for c in classes)))
    # the two hot bodies: the innermost loop of depth 4 (check pass, four edges per trip) and every block between the second and
    # third s_barrier of the iteration loop (bit pass: one round of T faults)
    inner4 = [r for r in rows if r[0]["inner"] and r[0]["depth"] == 4]
    summary = {}
    if inner4:
        c = inner4[0][1]
        summary["check_pass_loop_4_edges"] = c
        print("\n# check pass, innermost loop (4 edges per trip): " + ", ".join("%s %d" % (k, v) for k, v in c.items() if v))
        print("#   per edge: %.2f fast + %.2f slow VALU = %.1f issue clk at 2 / 4 clk" % (c["valu_fast"] / 4, c["valu_slow"] / 4,
              (2 * c["valu_fast"] + 4 * c["valu_slow"]) / 4))
    # bit pass: the depth-2 blocks that issue the 16-byte check-state gathers (ds_read_b128 through inline asm)
    bg = [cnt for b, cnt in rows if b["depth"] == 2 and not b["inner"] and any(i.startswith("ds_read_b128") for i in b["ins"])]
    if bg:
        g = sum(c["lds"] for c in bg)
        tot = {k: sum(c[k] for c in bg) for k in classes}
        summary["bit_pass_gather_blocks"] = tot
        summary["bit_pass_gathers"] = g
        print("# bit pass, the %d blocks holding the %d check-state gathers of a fault: " % (len(bg), g) + ", ".join("%s %d" % (k, v) for k, v in tot.items() if v))
        print("#   per edge: %.2f fast + %.2f slow VALU = %.1f issue clk at 2 / 4 clk" % (tot["valu_fast"] / g, tot["valu_slow"] / g,
              (2 * tot["valu_fast"] + 4 * tot["valu_slow"]) / g))
    # scatter kernel (bp_scatter.hip): the gather pass's innermost loop (four ds_read_b32 per trip) and the scatter pass's loop
    # (ds_add_u32: two groups of four edges per trip, every block of that loop counted)
    gl = [(sum(i.startswith("ds_read_b32") for i in b["ins"]), cnt) for b, cnt in rows if b["inner"]
          and sum(i.startswith("ds_read_b32") for i in b["ins"]) >= 4 and not any(i.startswith("ds_add_u32") for i in b["ins"])]
    if gl and "scatter" in a.kernel:
        ne, c8 = max(gl, key=lambda t: t[0])                  # the main loop: two groups of four edges per trip
        c = {k: v * 4.0 / ne for k, v in c8.items()}          # ... normalised to four edges, the unit bench.py prices
        summary["gather_pass_loop_4_edges"] = c
        print("\n# gather pass, innermost loop (%d edges per trip; counts per 4 edges): " % ne + ", ".join("%s %.1f" % (k, v) for k, v in c.items() if v))
        print("#   per edge: %.2f fast + %.2f slow VALU = %.1f issue clk at 2 / 4 clk" % (c["valu_fast"] / 4, c["valu_slow"] / 4,
              (2 * c["valu_fast"] + 4 * c["valu_slow"]) / 4))
        # the scatter pass's plain path: the block that builds three of a group's four values back to back (v_bitop3 0x78 =
        # pdif ^ (pxq & differ-mask), no tail predicate); the fourth is computed in the loop header, the same five instructions
        sl = [(b, cnt) for b, cnt in rows if sum("bitop3:0x78" in i for i in b["ins"]) == 3 and not any(i.startswith("v_cndmask") for i in b["ins"])]
        if sl:
            b, cnt = sl[0]
            summary["scatter_pass_plain_block"] = cnt
            summary["scatter_pass_edges_in_block"] = 3
            print("# scatter pass, plain block (3 of a group's 4 edges): " + ", ".join("%s %d" % (k, v) for k, v in cnt.items() if v))
            print("#   per edge: %.2f fast + %.2f slow VALU = %.1f issue clk at 2 / 4 clk" % (cnt["valu_fast"] / 3, cnt["valu_slow"] / 3,
                  (2 * cnt["valu_fast"] + 4 * cnt["valu_slow"]) / 3))
    mn = {}
    for b, cnt in rows:
        for ins in b["ins"]:
            if classify(ins).startswith("valu"):
                mn[ins.split()[0]] = mn.get(ins.split()[0], 0) + 1
    print("\n# VALU mnemonics of the whole kernel (static count): " + ", ".join("%s %d" % kv for kv in sorted(mn.items(), key=lambda kv: -kv[1])))
    if a.json:
        json.dump({"kernel": a.kernel, "summary": summary,
                   "blocks": [{"label": b["label"], "depth": b["depth"], "inner": b["inner"], **cnt} for b, cnt in rows]},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
