#!/usr/bin/env python3
"""Pins the oracle (oracle/qd_oracle.c, the CPU restatement of ldpc's BP / OSD) against `ldpc` ITSELF, the day the wheel exists.

The reference delegates all decoder arithmetic to the external `ldpc` package (pyproject.toml:35 `ldpc>=2.1.2`; call sites
decoder/sliding_window.py:149,171), which is neither under /root/reference nor installable in the build container or on the GPU box
(profiles/r02_probe_ldpc_stim.txt).  SURVEY.md Appendix A therefore restated its published algorithm from memory and tagged what could
not be checked as *verify*.  This script is the check: where `import ldpc` works it decodes the same syndromes with
ldpc.bposd_decoder.BpOsdDecoder and with the oracle in double precision on the exact channel LLRs, shot by shot, for every
(bp_method, schedule, osd_method) triple on the window fixtures, and reports

  V1  serial schedule: does ldpc's default `random_schedule_seed` leave the NATURAL fault order (the reference wrapper's default
      schedule is 'serial', decoder/bposd.py:54)?
  V2  `ms_scaling_factor` default (restated as 1.0; 0 = the 1 - 2^-t ramp)
  V3  `channel_probs` accepted as an alias of `error_channel` (the keyword the reference passes, decoder/bposd.py:83)
  V4  OSD-CS / OSD-E candidate cost = sum of log(1/p_j) over the flipped faults (soft weight, not Hamming), strict `<`
  V5  column order of OSD = ascending posterior LLR, ties by fault index

and writes tests/golden/ldpc_pin.npz (syndromes + ldpc's outputs) so that the pin travels to machines without the wheel
(tests/test_oracle.py::test_oracle_matches_ldpc_pin picks the file up when present).

Without ldpc: prints one line and exits 0."""
import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

FIXTURES = [                     # (fixture, window) -- the five window shapes the parity tests use
    ("hgp225_cardinal_r3_p0.01", None),            # configs[0], whole history 540 x 5409
    ("bb72_custom_r6_p0.003", None),               # configs[1], 288 x 2592
    ("bb72_custom_r6_p0.003", (3, 1, 0)),          # its W=3 F=1 window
    ("bb144_custom_r12_p0.003", None),             # configs[2], 1008 x 9504
    ("bb144_custom_r12_p0.003", (5, 3, 0)),        # W=5 F=3 window 360 x 3600 (the "reference settings" entry of bench.py)
]
TRIPLES = list(itertools.product(("minimum_sum", "product_sum"), ("parallel", "serial"), ("osd_0", "osd_cs", "osd_e")))


def window_of(name, win):
    import helpers
    if win is None:
        H, _, pri = helpers.dem_matrices(name)
        return H, pri
    W, F, k = win
    w = helpers.window_set(name, W, F)[k]
    return w["H"], w["priors"]


def ldpc_decode(cls, H, pri, synd, kw):
    """ldpc on every row of `synd`: (corrections, iterations, converged, bp decisions, posteriors)."""
    from scipy.sparse import csr_matrix
    dec = cls(csr_matrix(H), error_channel=list(np.asarray(pri, dtype=np.float64)), **kw)
    n = H.shape[1]
    out = np.zeros((len(synd), n), np.uint8)
    it = np.zeros(len(synd), np.int32)
    conv = np.zeros(len(synd), np.uint8)
    bpd = np.zeros((len(synd), n), np.uint8)
    llr = np.zeros((len(synd), n), np.float64)
    for i, s in enumerate(synd):
        out[i] = np.asarray(dec.decode(np.asarray(s, dtype=np.uint8)), dtype=np.uint8)
        it[i] = int(getattr(dec, "iter", -1))
        conv[i] = int(bool(getattr(dec, "converge", False)))
        if hasattr(dec, "bp_decoding"):
            bpd[i] = np.asarray(dec.bp_decoding, dtype=np.uint8)
        if hasattr(dec, "log_prob_ratios"):
            llr[i] = np.asarray(dec.log_prob_ratios, dtype=np.float64)
    return out, it, conv, bpd, llr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shots", type=int, default=96, help="syndromes per (fixture, triple)")
    ap.add_argument("--max-iter", type=int, default=12)
    ap.add_argument("--osd-order", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ldpc_pin.npz"))
    args = ap.parse_args()

    import refhooks
    cls, info = refhooks.probe_ldpc()
    if cls is None:
        print("ldpc not importable (%s): nothing to pin; the oracle stays 'parity unpinned' for the BP/OSD arithmetic" % info)
        return 0
    import oracle as orc

    report = {"ldpc_version": info, "max_iter": args.max_iter, "osd_order": args.osd_order, "cases": [], "verify": {}}
    store = {}
    agree = {"decisions": 0, "iters": 0, "total": 0}
    per_v = {"V1_serial_natural_order": [0, 0], "V4_osd_soft_cost": [0, 0], "V5_column_order": [0, 0]}
    for (name, win) in FIXTURES:
        H, pri = window_of(name, win)
        tag = name + ("" if win is None else "_W%dF%d_k%d" % win)
        rng = np.random.default_rng(2026)
        errs = (rng.random((args.shots, H.shape[1])) < np.asarray(pri)[None, :]).astype(np.uint8)
        synd = (errs @ H.T.toarray().astype(np.uint8) % 2).astype(np.uint8) if H.shape[1] < 6000 else \
            np.asarray((H.tocsr() @ errs.T.astype(np.int32)).T % 2, dtype=np.uint8)
        store[tag + "/syndromes"] = np.packbits(synd, axis=1)
        g = orc.Graph(H, pri)
        for (bpm, sch, osd) in TRIPLES:
            order = 0 if osd == "osd_0" else args.osd_order
            kw = dict(max_iter=args.max_iter, bp_method=bpm, schedule=sch, osd_method=osd, osd_order=order)
            out, it, conv, bpd, llr = ldpc_decode(cls, H, pri, synd, kw)
            prm = orc.make_params(bpm, sch, args.max_iter, osd, order, 1.0, orc.FORM_LDPC_F64)
            ref, flags = g.decode_batch(synd, prm)
            same = (ref == out).all(axis=1)
            # flags: [converged, iterations, pivots, inconsistent] (oracle/qd_oracle.c oq_bposd_decode)
            same_it = (flags[:, 1] == it) | (it < 0)
            agree["decisions"] += int(same.sum()); agree["iters"] += int(same_it.sum()); agree["total"] += len(synd)
            if sch == "serial":
                per_v["V1_serial_natural_order"][0] += int(same.sum()); per_v["V1_serial_natural_order"][1] += len(synd)
            if osd != "osd_0":
                nc = conv == 0
                per_v["V4_osd_soft_cost"][0] += int(same[nc].sum()); per_v["V4_osd_soft_cost"][1] += int(nc.sum())
            else:
                nc = conv == 0
                per_v["V5_column_order"][0] += int(same[nc].sum()); per_v["V5_column_order"][1] += int(nc.sum())
            key = "%s/%s_%s_%s" % (tag, bpm, sch, osd)
            store[key + "/out"] = np.packbits(out, axis=1)
            store[key + "/iter"] = it
            store[key + "/converged"] = conv
            report["cases"].append({"case": key, "shots": len(synd), "identical_corrections": int(same.sum()),
                                    "identical_iterations": int(same_it.sum()), "bp_converged": int(conv.sum())})
            print("%-64s corrections %3d/%3d  iterations %3d/%3d  (BP converged on %d)" % (key, same.sum(), len(synd), same_it.sum(), len(synd), conv.sum()))
    # V2 / V3: constructor defaults and the alias
    from scipy.sparse import csr_matrix
    H, pri = window_of("bb72_custom_r6_p0.003", (3, 1, 0))
    d = cls(csr_matrix(H), error_channel=list(pri), max_iter=4, bp_method="minimum_sum")
    report["verify"]["V2_ms_scaling_factor_default"] = float(getattr(d, "ms_scaling_factor", float("nan")))
    try:
        d2 = cls(csr_matrix(H), channel_probs=list(pri), max_iter=4, bp_method="minimum_sum")
        s = np.zeros(H.shape[0], np.uint8); s[:3] = 1
        report["verify"]["V3_channel_probs_alias"] = bool(np.array_equal(np.asarray(d.decode(s)), np.asarray(d2.decode(s))))
    except Exception as exc:
        report["verify"]["V3_channel_probs_alias"] = "rejected: %s" % exc
    for k, (a, b) in per_v.items():
        report["verify"][k] = {"identical": a, "of": b, "holds": bool(b > 0 and a == b)}
    report["summary"] = agree
    store["report_json"] = np.frombuffer(json.dumps(report).encode(), dtype=np.uint8)
    np.savez_compressed(args.out, **store)
    print(json.dumps(report["verify"], indent=1))
    print("identical corrections %d / %d, identical iteration counts %d / %d -> %s" % (agree["decisions"], agree["total"], agree["iters"], agree["total"], args.out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
