#!/usr/bin/env python3
"""Anchor the whole path -- DEM extractor, product-sum / serial BP, OSD-CS, BP-LSD, the sliding-window drivers -- to the only
outputs of the REAL ldpc + Stim pipeline this environment holds: the printed results of the reference's executed notebook
cells.  Every case below repeats one cell's call (same code, circuit generator, rounds, p, W, F, decoder keywords) on DEM-sampled
shots and asks whether the failure probability measured here is compatible with the published count k of n trials, i.e. lies
inside the exact (Clopper-Pearson) 95 % interval of k / n.

    python tools/published_anchor.py [--shots 131072] [--oracle-shots 4096] [--out profiles/r03_published_anchor.json]

For each case:
  device        pL over `--shots` shots through the public API on the MI355X
  oracle_f64    pL of the CPU oracle in ldpc's arithmetic (double, exact LLRs) on the first `--oracle-shots` of the same shots
  device == f32 mirror on those shots?   (the device's own arithmetic, bit for bit)
so that a miss can be attributed: DEM priors / extractor (oracle_f64 misses too), float product-sum (device differs from
oracle_f64 beyond sampling error), or the device kernels (device differs from its mirror).

Fixtures: tests/golden/circuits/*.stim.gz were emitted by the reference's own circuit generators (tools/gen_fixtures.py anchor);
nothing here reads /root/reference.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import helpers  # noqa: E402

# (id, notebook cell, circuit fixture, code, rounds, p, W, F, kind, decoder keywords, published failures, published trials)
_BPOSD = dict(max_iter=10, osd_order=1)        # "max_iter, osd_order = 10, 1"; every other keyword is the wrapper's default
CASES = [
    ("06A_hgp_p5e-4", "doc/06A_end_to_end_demo_hgp.ipynb cell 5", "hgp225_cardinal_r15_p0.001", "hgp225", 15, 5e-4, 5, 3, "circuit", _BPOSD, 0, 200),
    ("06A_hgp_p1e-3", "doc/06A_end_to_end_demo_hgp.ipynb cell 5", "hgp225_cardinal_r15_p0.001", "hgp225", 15, 1e-3, 5, 3, "circuit", _BPOSD, 1, 200),
    ("06A_hgp_p2e-3", "doc/06A_end_to_end_demo_hgp.ipynb cell 5", "hgp225_cardinal_r15_p0.001", "hgp225", 15, 2e-3, 5, 3, "circuit", _BPOSD, 24, 200),
    ("06B_bb90_p5e-4", "doc/06B_end_to_end_demo_bb.ipynb cell 5", "bb90_custom_r15_p0.001", "bb90", 15, 5e-4, 5, 3, "circuit", _BPOSD, 0, 1000),
    ("06B_bb90_p1e-3", "doc/06B_end_to_end_demo_bb.ipynb cell 5", "bb90_custom_r15_p0.001", "bb90", 15, 1e-3, 5, 3, "circuit", _BPOSD, 1, 1000),
    ("06B_bb90_p2e-3", "doc/06B_end_to_end_demo_bb.ipynb cell 5", "bb90_custom_r15_p0.001", "bb90", 15, 2e-3, 5, 3, "circuit", _BPOSD, 2, 1000),
    ("04_hgp_circuit", "doc/04_decoding_sliding_window.ipynb cell 9", "hgp225_cardinal_r15_p0.001", "hgp225", 15, 1e-3, 5, 3, "circuit", _BPOSD, 1, 100),
    ("04_hgp_phenom", "doc/04_decoding_sliding_window.ipynb cell 8", "hgp225_cardinal_r15_p0.001", "hgp225", 15, 1e-3, 5, 3, "phenom",
     dict(_BPOSD, eff_error_rate_per_fault=0.011), 4, 100),
    ("05_hgp_phenom", "doc/05_decoder_variants.ipynb cell 8 (same call as 04 cell 8, another execution)", "hgp225_cardinal_r15_p0.001", "hgp225", 15, 1e-3, 5, 3,
     "phenom", dict(_BPOSD, eff_error_rate_per_fault=0.011), 6, 100),
    ("05_hgp_phenom_bplsd", "doc/05_decoder_variants.ipynb cell 9", "hgp225_cardinal_r15_p0.001", "hgp225", 15, 1e-3, 5, 3, "phenom_lsd",
     dict(max_iter=10, lsd_order=1, eff_error_rate_per_fault=0.011), 20, 100),
    ("00_hgprep3", "doc/00_getting_started.ipynb cell 8", "hgprep3_zxcoloration_r3_p0.001", "hgp_rep3", 3, 1e-3, 3, 2, "circuit", _BPOSD, 0, 100),
]


def clopper_pearson(k, n, conf=0.95):
    from scipy.stats import beta
    a = (1.0 - conf) / 2.0
    lo = 0.0 if k == 0 else float(beta.ppf(a, k, n - k + 1))
    hi = 1.0 if k == n else float(beta.ppf(1.0 - a, k + 1, n - k))
    return lo, hi


def circuit_for(case):
    _, _, fixture, _, _, p, *_ = case
    return helpers.circuit_text(fixture) if abs(p - 1e-3) < 1e-12 else helpers.circuit_text_at_p(fixture, 1e-3, p)


def device_decode(case, det):
    from quits_amd import decoder as qd
    from quits_amd.dem import Circuit
    _, _, _, code, _, _, W, F, kind, kw, _, _ = case
    cd = helpers.code(code)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == "circuit":
            return qd.sliding_window_bposd_circuit_mem(det, Circuit(circuit_for(case)), cd["hz"], cd["lz"], W, F, **kw)
        if kind == "phenom":
            return qd.sliding_window_bposd_phenom_mem(det, cd["hz"], cd["lz"], W, F, **kw)
        return qd.sliding_window_bplsd_phenom_mem(det, cd["hz"], cd["lz"], W, F, **kw)


def oracle_windows(case):
    """Window list in the oracle's format + its decoder parameters (wrapper defaults: product_sum, serial, osd_cs / lsd_cs)."""
    import oracle as orc
    from quits_amd.decoder.base import spacetime, window_count
    from quits_amd.decoder.sliding_window import phenom_window_set
    from quits_amd.dem import Circuit
    _, _, _, code, R, _, W, F, kind, kw, _, _ = case
    cd = helpers.code(code)
    nz = cd["hz"].shape[0]
    ncr, W_last, _ = window_count(R, W, F)
    if kind == "circuit":
        a, b, pp, d = spacetime(Circuit(circuit_for(case)), cd["hz"], W, F, ncr)
    else:
        a, b, pp, d = phenom_window_set(cd["hz"], cd["lz"], W, F, R, kw["eff_error_rate_per_fault"], kw["eff_error_rate_per_fault"])
    wins = [{"H": a[k], "L": b[k], "priors": pp[k], "U": d[k] if k < len(d) else None, "row0": F * k * nz} for k in range(len(a))]
    if kind == "phenom_lsd":
        prm = ("product_sum", "serial", kw["max_iter"], "lsd_cs", kw["lsd_order"])
    else:
        prm = ("product_sum", "serial", kw["max_iter"], "osd_cs", kw["osd_order"])
    return wins, nz, prm


def _oracle_worker(args):
    import oracle as orc
    case, det, form = args
    wins, nz, prm = oracle_windows(case)
    pred, _ = orc.sliding_window_decode(wins, nz, det, orc.make_params(*prm, 1.0, form))
    return pred


def oracle_decode(case, det, form, pool, nproc):
    parts = np.array_split(det, nproc)
    res = pool.map(_oracle_worker, [(case, part, form) for part in parts if len(part)])
    return np.concatenate(res, axis=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shots", type=int, default=1 << 17)
    ap.add_argument("--oracle-shots", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_published_anchor.json"))
    a = ap.parse_args()
    import torch
    import oracle as orc
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.decoder.device import DemSampler
    from quits_amd.dem import Circuit
    nproc = len(os.sched_getaffinity(0))
    pool = mp.get_context("fork").Pool(nproc) if a.oracle_shots > 0 else None
    rows = []
    for case in CASES:
        cid, cell, fixture, code, R, p, W, F, kind, kw, k_pub, n_pub = case
        if a.only and a.only not in cid:
            continue
        H, L, pri = detector_error_model_to_matrix(Circuit(circuit_for(case)).detector_error_model())
        det, obs = DemSampler(H, L, pri).sample(a.shots, seed=a.seed)
        torch.cuda.synchronize()
        t0 = time.time()
        try:
            pred = device_decode(case, det)
        except NotImplementedError as exc:
            rows.append(dict(case=cid, cell=cell, error="NotImplementedError: %s" % exc))
            print(rows[-1], flush=True)
            continue
        t_dev = time.time() - t0
        obs_h = obs.cpu().numpy()
        fail_dev = (pred != obs_h).any(axis=1)
        lo, hi = clopper_pearson(k_pub, n_pub)
        pl = float(fail_dev.mean())
        row = dict(case=cid, cell=cell, code=code, rounds=R, p=p, W=W, F=F, kind=kind, kwargs=kw,
                   published=dict(failures=k_pub, trials=n_pub, pL=k_pub / n_pub, cp95=[lo, hi]),
                   device=dict(shots=a.shots, failures=int(fail_dev.sum()), pL=pl,
                               sigma=float(np.sqrt(max(pl * (1 - pl), 1e-12) / a.shots)), seconds=round(t_dev, 2),
                               inside_cp95=bool(lo <= pl <= hi)),
                   dem=dict(detectors=int(H.shape[0]), faults=int(H.shape[1]), sum_priors=float(pri.sum())))
        if pool is not None:
            n_or = min(a.oracle_shots, a.shots)
            det_h = det[:n_or].cpu().numpy()
            t0 = time.time()
            p64 = oracle_decode(case, det_h, orc.FORM_LDPC_F64, pool, nproc)
            t64 = time.time() - t0
            p32 = oracle_decode(case, det_h, orc.FORM_LDPC_F32, pool, nproc)
            f64 = (p64 != obs_h[:n_or]).any(axis=1)
            fd = fail_dev[:n_or]
            b, c = int((fd & ~f64).sum()), int((~fd & f64).sum())
            row["oracle_f64"] = dict(shots=n_or, failures=int(f64.sum()), pL=float(f64.mean()), inside_cp95=bool(lo <= f64.mean() <= hi),
                                     seconds=round(t64, 1), cores=nproc,
                                     device_failures_same_shots=int(fd.sum()), discordant=[b, c],
                                     mcnemar_z=float((b - c) / np.sqrt(b + c)) if b + c else 0.0)
            row["device_equals_f32_mirror"] = dict(shots=n_or, identical=bool(np.array_equal(pred[:n_or], p32.astype(np.int64))),
                                                   differing_shots=int((pred[:n_or] != p32).any(axis=1).sum()))
        rows.append(row)
        print(json.dumps(row), flush=True)
    if pool is not None:
        pool.close()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(note=__doc__.strip().split("\n\n")[0], seed=a.seed, cases=rows), open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
