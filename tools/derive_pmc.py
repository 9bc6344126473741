#!/usr/bin/env python3
"""profiles/pmc_traffic.json entries from one evidence set (tools/r04_evidence.sh <tag> -> gpurun_out/):
  python tools/derive_pmc.py <tag> <profiles-prefix>      e.g.  python tools/derive_pmc.py r04 r04
HBM bytes per launch = FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, separate rocprofv3 --pmc passes, largest
dispatch of the kernel; SQ ratios from the single-launch counter passes (tools/pmc_bp_kernel.sh, tools/pmc_osd_kernel.sh)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")


def pmc_rows(path, counter):
    out, on = {}, False
    for ln in open(path):
        if ln.startswith("# %s per kernel" % counter):
            on = True
            continue
        if ln.startswith("#"):
            on = False
        if on and ln.strip():
            f = ln.split()
            out[" ".join(f[:-4])] = float(f[-1])          # largest dispatch, KiB
    return out


def sq(path, kernel):
    d, on = {}, False
    for ln in open(path):
        if ln.startswith("== "):
            on = ln[3:].startswith(kernel)
            continue
        if on and ln.strip() and not ln.startswith("#"):
            f = ln.split()
            d[f[0]] = float(f[1])
    return d


summ = os.path.join(G, "prof_%s" % tag, "summary.txt")
fetch, write = pmc_rows(summ, "FETCH_SIZE"), pmc_rows(summ, "WRITE_SIZE")
pm_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
pm = json.load(open(pm_path))
key = "p0.003_it50_W14_F1_shots65536"


def find(d, sub):
    return next(v for k, v in d.items() if sub in k)


def ratios(c, src):
    return {"valu_insts_per_cu_clk": round(c["SQ_INSTS_VALU"] / c["SQ_BUSY_CU_CYCLES"], 4),
            "frac_of_2_per_cu_clk": round(c["SQ_INSTS_VALU"] / c["SQ_BUSY_CU_CYCLES"] / 2.0, 4),
            "salu_per_valu": round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3),
            "wait_any_over_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
            "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3), "source": src}


bpk = "qd_bp_scatter_wide_kernel"
f_kib, w_kib = find(fetch, bpk), find(write, bpk)
pm[key] = {"kernel": bpk, "bp_bytes_per_launch": int(f_kib * 2048 + w_kib * 1024),
           "source": "profiles/%s_rocprofv3_summary.txt: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction of MI355X_MICROARCH.md) and --pmc WRITE_SIZE, "
                     "separate passes; largest dispatch of %s<512, 8, 2, 2> (the round's final binary)" % (pre, bpk),
           "fetch_size_kib": f_kib, "write_size_kib": w_kib,
           "sq_counters": ratios(sq(os.path.join(G, "pmcbp_%s" % tag, "summary.txt"), bpk),
                                 "profiles/%s_pmc_sq_bp_kernel.txt (tools/pmc_bp_kernel.sh on the final binary: SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES)" % pre)}
ok = "qd_osd0_sr_kernel"
f_kib, w_kib = find(fetch, ok), find(write, ok)
chain = {}
tm = os.path.join(G, tag, "osd_phase_timers.txt")
if os.path.exists(tm):
    m = re.search(r"per shot: rounds ([\d.]+) tiers ([\d.]+) batches ([\d.]+)", open(tm).read())
    if m:
        chain = {"barrier_rounds_per_shot": float(m.group(1)), "tiers_per_shot": float(m.group(2)), "batches_per_shot": float(m.group(3)),
                 "source": "profiles/%s_osd_phase_timers.txt (tools/osd_timing.py, QD_OSD_TIMING build; two barriers per round, two per batch set-up)" % pre}
pm["osd_" + key] = {"kernel": ok, "bytes_per_launch": int(f_kib * 2048 + w_kib * 1024), "fetch_size_kib": f_kib, "write_size_kib": w_kib,
                    "source": "profiles/%s_rocprofv3_summary.txt: FETCH_SIZE x 2 + WRITE_SIZE, separate passes; largest dispatch of %s<512, 2, 4>" % (pre, ok),
                    "sq_counters": ratios(sq(os.path.join(G, "pmcosd_%s" % tag, "summary.txt"), ok),
                                          "profiles/%s_pmc_sq_osd_kernels.txt (tools/pmc_osd_kernel.sh: one launch over the failing shots of 32 768)" % pre),
                    "chain": chain}
json.dump(pm, open(pm_path, "w"), indent=1)
print(json.dumps({k: pm[k] for k in (key, "osd_" + key)}, indent=1))
