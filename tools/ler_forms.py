#!/usr/bin/env python3
"""Which part of the arithmetic moves the logical error rate?  

Decodes the SAME Philox-sampled syndromes of BASELINE's headline configuration (BB [[144,12,12]], R = 12, p = 0.003,
single window, min-sum flooding, max_iter 50, ms_scaling 1.0, OSD-0) with every arithmetic form of the CPU oracle and
reports the PAIRED comparison (McNemar) of each form against the double-precision / ldpc-update-order form, the
stand-in for ldpc.BpOsdDecoder:

  ldpc_f64      per-edge messages, double, ldpc's prefix/suffix sums            (reference arithmetic)
  comp_f64      compressed check state + "total minus own", double
  ldpc_f32      per-edge messages, float                                         (= bp_general.hip, bit for bit)
  comp_f32      compressed, float                                                (= round-1 bp_kernels.hip, bit for bit)
  ldpc_f64_qK   reference arithmetic on channel LLRs rounded to multiples of 2^-K: every form then computes exactly
                (no rounding anywhere), so this one column stands for all four forms AND for the round-2 HIP kernel

  python tools/ler_forms.py run <shots> <seed> <out.npz> [procs] [configs,comma,separated]
  python tools/ler_forms.py report <out.npz> [<out2.npz> ...]          # pooled over the files given
CPU only (oracle); run it where the cores are (the GPU box has 256 hardware threads)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

CHUNK, NAME, MAX_ITER = 2000, "bb144_custom_r12_p0.003", 50
CONFIGS = {  # name -> (oracle form, LLR fraction bits or -1)
    "ldpc_f64": (0, -1), "comp_f64": (2, -1), "ldpc_f32": (3, -1), "comp_f32": (1, -1),
    "ldpc_f64_q9": (0, 9), "ldpc_f64_q11": (0, 11), "ldpc_f64_q12": (0, 12), "ldpc_f64_q16": (0, 16), "ldpc_f64_q20": (0, 20),
    "comp_f32_q16": (1, 16),
}
_G = {}


def _setup(names):
    import helpers, oracle as orc
    H, L, pri = helpers.dem_matrices(NAME)
    from scipy.sparse import csr_matrix
    _G.update(H=H, L=L, pri=pri, Lc=csr_matrix(L, dtype=np.int32), orc=orc, names=names,
              graphs={nm: orc.Graph(H, pri).quantize_llr(CONFIGS[nm][1]) for nm in names},
              w=(1 << np.arange(L.shape[0])).astype(np.int64))


def _chunk(args):
    c, seed = args
    orc = _G["orc"]
    det, obs, _ = orc.sample_dem(_G["H"], _G["L"], _G["pri"], seed=seed, shot0=c * CHUNK, B=CHUNK)
    out = {"obs": (obs.astype(np.int64) @ _G["w"]).astype(np.uint16)}
    for nm in _G["names"]:
        prm = orc.make_params("minimum_sum", "parallel", MAX_ITER, "osd_0", 0, 1.0, CONFIGS[nm][0])
        orc.max_abs_llr(True)
        err, flags = _G["graphs"][nm].decode_batch(det, prm)
        pred = np.asarray((_G["Lc"] @ err.T.astype(np.int32)) % 2).T
        out[nm] = ((pred.astype(np.int64) @ _G["w"]).astype(np.uint16), flags[:, 0].astype(np.uint8),
                   flags[:, 1].astype(np.uint8), orc.max_abs_llr())
    return c, out


def run(shots, seed, path, procs, names):
    import multiprocessing as mp
    nch = shots // CHUNK
    t0 = time.time()
    with mp.Pool(procs, initializer=_setup, initargs=(names,)) as pool:
        res = sorted(pool.imap_unordered(_chunk, [(c, seed) for c in range(nch)], chunksize=1), key=lambda r: r[0])
    arrs = {"obs": np.concatenate([r[1]["obs"] for r in res])}
    meta = {"config": NAME, "max_iter": MAX_ITER, "seed": seed, "shots": nch * CHUNK, "procs": procs,
            "seconds": time.time() - t0, "names": names, "max_abs_llr": {}}
    for nm in names:
        arrs[nm + "_pred"] = np.concatenate([r[1][nm][0] for r in res])
        arrs[nm + "_conv"] = np.packbits(np.concatenate([r[1][nm][1] for r in res]))
        arrs[nm + "_iters"] = np.concatenate([r[1][nm][2] for r in res])
        meta["max_abs_llr"][nm] = max(r[1][nm][3] for r in res)
    arrs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.0f s" % meta["seconds"])


def report(paths):
    from math import erfc, sqrt
    zs = [np.load(p) for p in paths]
    metas = [json.loads(bytes(z["meta"]).decode()) for z in zs]
    names = [nm for nm in metas[0]["names"] if all(nm in m["names"] for m in metas)]
    obs = np.concatenate([z["obs"] for z in zs])
    N = len(obs)
    fail = {nm: np.concatenate([z[nm + "_pred"] for z in zs]) != obs for nm in names}
    pred = {nm: np.concatenate([z[nm + "_pred"] for z in zs]) for nm in names}
    iters = {nm: np.concatenate([z[nm + "_iters"] for z in zs]) for nm in names}
    ref = "ldpc_f64"
    out = {"shots": N, "files": [os.path.basename(p) for p in paths], "seeds": [m["seed"] for m in metas],
           "config": metas[0]["config"] + ", min-sum flooding max_iter=%d ms_scaling=1.0 + OSD-0" % MAX_ITER,
           "reference_form": ref, "forms": {}}
    pr = fail[ref].mean()
    for nm in names:
        p = fail[nm].mean()
        rec = {"fails": int(fail[nm].sum()), "pL": p, "sigma_unpaired": sqrt(p * (1 - p) / N),
               "max_abs_llr": max(m["max_abs_llr"][nm] for m in metas), "mean_iters": float(iters[nm].mean())}
        if nm != ref:
            b = int((fail[nm] & ~fail[ref]).sum())      # this form fails, the reference form does not
            c = int((~fail[nm] & fail[ref]).sum())
            rec.update({"identical_predictions": float((pred[nm] == pred[ref]).mean()),
                        "identical_iteration_counts": float((iters[nm] == iters[ref]).mean()),
                        "only_this_fails": b, "only_reference_fails": c, "delta_pL": p - pr,
                        "delta_in_unpaired_sigma": (p - pr) / sqrt(pr * (1 - pr) / N),
                        "mcnemar_z": (b - c) / sqrt(b + c) if b + c else 0.0,
                        "mcnemar_p_two_sided": erfc(abs(b - c) / sqrt(2.0 * (b + c))) if b + c else 1.0})
        out["forms"][nm] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        names = sys.argv[6].split(",") if len(sys.argv) > 6 else list(CONFIGS)
        run(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else os.cpu_count(), names)
    else:
        report(sys.argv[2:])
