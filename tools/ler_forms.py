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

  k1_grid / k1g_grid / k1_raw   (gpu mode) the HIP library itself: LDS kernel on its LLR grid (the product path), the
                one-message-per-edge kernel on the same grid, and the LDS kernel with QD_FLAG_RAW_LLR (round-1 arithmetic)

  python tools/ler_forms.py run <shots> <seed> <out.npz> [procs] [configs,comma,separated] [first shot]   # CPU oracle
  python tools/ler_forms.py gpu <shots> <seed> <out.npz>                                        # on the GPU box
  python tools/ler_forms.py report <out.npz> [<out2.npz> ...]   # files of one seed are merged, seeds are pooled
The GPU box grants 16 CPUs (cgroup), this container 8: a 10^6-shot double-precision column takes ~20 core-minutes."""
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
    c, seed, shot0 = args
    orc = _G["orc"]
    det, obs, _ = orc.sample_dem(_G["H"], _G["L"], _G["pri"], seed=seed, shot0=shot0 + c * CHUNK, B=CHUNK)
    out = {"obs": (obs.astype(np.int64) @ _G["w"]).astype(np.uint16)}
    for nm in _G["names"]:
        prm = orc.make_params("minimum_sum", "parallel", MAX_ITER, "osd_0", 0, 1.0, CONFIGS[nm][0])
        orc.max_abs_llr(True)
        err, flags = _G["graphs"][nm].decode_batch(det, prm)
        pred = np.asarray((_G["Lc"] @ err.T.astype(np.int32)) % 2).T
        out[nm] = ((pred.astype(np.int64) @ _G["w"]).astype(np.uint16), flags[:, 0].astype(np.uint8),
                   flags[:, 1].astype(np.uint8), orc.max_abs_llr())
    return c, out


def run(shots, seed, path, procs, names, shot0=0):
    import multiprocessing as mp
    nch = shots // CHUNK
    t0 = time.time()
    with mp.Pool(procs, initializer=_setup, initargs=(names,)) as pool:
        res = sorted(pool.imap_unordered(_chunk, [(c, seed, shot0) for c in range(nch)], chunksize=1), key=lambda r: r[0])
    arrs = {"obs": np.concatenate([r[1]["obs"] for r in res])}
    meta = {"config": NAME, "max_iter": MAX_ITER, "seed": seed, "shot0": shot0, "shots": nch * CHUNK, "procs": procs,
            "seconds": time.time() - t0, "names": names, "max_abs_llr": {}}
    for nm in names:
        arrs[nm + "_pred"] = np.concatenate([r[1][nm][0] for r in res])
        arrs[nm + "_conv"] = np.packbits(np.concatenate([r[1][nm][1] for r in res]))
        arrs[nm + "_iters"] = np.concatenate([r[1][nm][2] for r in res])
        meta["max_abs_llr"][nm] = max(r[1][nm][3] for r in res)
    arrs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.0f s" % meta["seconds"])


def gpu(shots, seed, path):
    import torch, helpers
    from quits_amd.decoder.device import BatchDecoder, DemSampler, GF2Matrix, WindowGraph
    H, L, pri = helpers.dem_matrices(NAME)
    smp, g, Lm = DemSampler(H, L, pri), WindowGraph(H, pri), GF2Matrix(L)
    decs = {"k1_grid": BatchDecoder(g, max_iter=MAX_ITER, osd_method="osd_0"),
            "k1g_grid": BatchDecoder(g, max_iter=MAX_ITER, osd_method="osd_0", edge_messages=True),
            "k1_raw": BatchDecoder(g, max_iter=MAX_ITER, osd_method="osd_0", raw_llr=True)}
    w = (1 << torch.arange(L.shape[0], device="cuda")).to(torch.int32)
    t0 = time.time()
    B = 50000
    acc = {nm: ([], [], []) for nm in decs}
    obs_all, coarse = [], 0
    for c in range(shots // B):
        det, obs = smp.sample(B, seed=seed, shot0=c * B)
        obs_all.append((obs.to(torch.int32) * w).sum(1).to(torch.int16).cpu().numpy().view(np.uint16))
        for nm, dec in decs.items():
            bits, status = dec.decode(det)
            pred = torch.zeros((B, L.shape[0]), dtype=torch.uint8, device="cuda")
            Lm.xor_apply(bits, pred, accumulate=False)
            acc[nm][0].append((pred.to(torch.int32) * w).sum(1).to(torch.int16).cpu().numpy().view(np.uint16))
            acc[nm][1].append(((status >> 16) & 1).to(torch.uint8).cpu().numpy())
            acc[nm][2].append((status & 0x3FFF).clamp(max=255).to(torch.uint8).cpu().numpy())
            if nm == "k1_grid":
                coarse += int(((status >> 14) & 3).ne(0).sum().item())
    arrs = {"obs": np.concatenate(obs_all)}
    meta = {"config": NAME, "max_iter": MAX_ITER, "seed": seed, "shots": (shots // B) * B, "seconds": time.time() - t0,
            "names": list(decs), "max_abs_llr": {nm: 0.0 for nm in decs}, "k1_grid_info": decs["k1_grid"].info(),
            "k1_grid_shots_off_the_fine_grid": coarse}
    for nm in decs:
        arrs[nm + "_pred"] = np.concatenate(acc[nm][0])
        arrs[nm + "_conv"] = np.packbits(np.concatenate(acc[nm][1]))
        arrs[nm + "_iters"] = np.concatenate(acc[nm][2])
    arrs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.0f s" % meta["seconds"], meta["k1_grid_info"], "off the fine grid:", coarse)


def report(paths):
    """Files are placed by (seed, first shot); a form counts on the shots every form of that seed covers."""
    from math import erfc, sqrt
    by_seed = {}
    for p in paths:
        z = np.load(p)
        m = json.loads(bytes(z["meta"]).decode())
        ent = by_seed.setdefault(m["seed"], {"pieces": [], "mx": {}, "extra": {}})
        ent["pieces"].append((m.get("shot0", 0), m["shots"], m["names"], z))
        for k, v in m["max_abs_llr"].items():
            ent["mx"][k] = max(ent["mx"].get(k, 0.0), v)
        ent["extra"].update({k: v for k, v in m.items() if k.startswith("k1_")})
    seeds = sorted(by_seed)
    cols = {}
    for sd in seeds:
        ent = by_seed[sd]
        n = max(a + b for a, b, _, _ in ent["pieces"])
        obs, have_obs = np.zeros(n, np.uint16), np.zeros(n, bool)
        arr = {}
        for a, b, names, z in ent["pieces"]:
            sl = slice(a, a + b)
            assert not have_obs[sl].any() or np.array_equal(obs[sl][have_obs[sl]], z["obs"][:b][have_obs[sl]]), "observables disagree"
            obs[sl] = z["obs"][:b]; have_obs[sl] = True
            for nm in names:
                pr_, it_, hv_ = arr.setdefault(nm, (np.zeros(n, np.uint16), np.zeros(n, np.uint8), np.zeros(n, bool)))
                pr_[sl] = z[nm + "_pred"][:b]; it_[sl] = z[nm + "_iters"][:b]; hv_[sl] = True
        ent["obs"], ent["arr"] = obs, arr
    names = [nm for nm in by_seed[seeds[0]]["arr"] if all(nm in by_seed[sd]["arr"] for sd in seeds)]
    obs_l, pred_l, it_l = [], {nm: [] for nm in names}, {nm: [] for nm in names}
    for sd in seeds:
        ent = by_seed[sd]
        ok = np.ones(len(ent["obs"]), bool)
        for nm in names:
            ok &= ent["arr"][nm][2]
        obs_l.append(ent["obs"][ok])
        for nm in names:
            pred_l[nm].append(ent["arr"][nm][0][ok]); it_l[nm].append(ent["arr"][nm][1][ok])
        ent["covered"] = int(ok.sum())
    obs = np.concatenate(obs_l)
    N = len(obs)
    pred = {nm: np.concatenate(pred_l[nm]) for nm in names}
    iters = {nm: np.concatenate(it_l[nm]) for nm in names}
    fail = {nm: pred[nm] != obs for nm in names}
    metas = [{"seed": sd, "max_abs_llr": by_seed[sd]["mx"]} for sd in seeds]
    ref = "ldpc_f64"
    out = {"shots": N, "shots_per_seed": {str(sd): by_seed[sd]["covered"] for sd in seeds}, "files": len(paths), "seeds": seeds,
           "config": NAME + ", min-sum flooding max_iter=%d ms_scaling=1.0 + OSD-0" % MAX_ITER,
           "reference_form": ref, "device": {str(sd): by_seed[sd]["extra"] for sd in seeds}, "forms": {}}
    pr = fail[ref].mean()
    for nm in names:
        p = fail[nm].mean()
        rec = {"fails": int(fail[nm].sum()), "pL": p, "sigma_unpaired": sqrt(p * (1 - p) / N),
               "max_abs_llr": max(m["max_abs_llr"].get(nm, 0.0) for m in metas), "mean_iters": float(iters[nm].mean())}
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
    if sys.argv[1] == "gpu":
        gpu(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    elif sys.argv[1] == "run":
        names = sys.argv[6].split(",") if len(sys.argv) > 6 else list(CONFIGS)
        run(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else os.cpu_count(), names,
            int(sys.argv[7]) if len(sys.argv) > 7 else 0)
    else:
        report(sys.argv[2:])
