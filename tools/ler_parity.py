#!/usr/bin/env python3
"""Logical error rate of the device decoder and of the CPU oracle (double precision, ldpc's update order) on the SAME
syndromes: BASELINE's headline configuration, DEM-sampled with the Philox sampler that is bit-identical on CPU and GPU.

  python tools/ler_parity.py cpu  <shots> <out.json> [procs]     # here (no GPU): oracle, shot ranges over processes
  python tools/ler_parity.py gpu|gpu-edge <shots> <out.json>     # on the GPU box (compressed LDS kernel | per-edge kernel)
  python tools/ler_parity.py cmp  <cpu.json> <gpu.json>
Chunks of 10000 shots, chunk c = global shots [c*10000, (c+1)*10000), seed 1; per-chunk failure counts are stored."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
CHUNK, SEED, NAME, MAX_ITER = 10000, 1, "bb144_custom_r12_p0.003", 50


def cpu_chunk(c):
    import helpers, oracle as orc
    H, L, pri = helpers.dem_matrices(NAME)
    det, obs, _ = orc.sample_dem(H, L, pri, seed=SEED, shot0=c * CHUNK, B=CHUNK)
    g = orc.Graph(H, pri)
    prm = orc.make_params("minimum_sum", "parallel", MAX_ITER, "osd_0", 0, 1.0, orc.FORM_LDPC_F64)
    err, flags = g.decode_batch(det, prm)
    from scipy.sparse import csr_matrix
    pred = np.asarray((csr_matrix(L, dtype=np.int32) @ err.T.astype(np.int32)) % 2).T
    return c, int((pred != obs).any(axis=1).sum()), int(flags[:, 0].sum())


def main():
    mode = sys.argv[1]
    if mode == "cmp":
        a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
        n = min(len(a["fails"]), len(b["fails"])) * CHUNK
        fa, fb = sum(a["fails"][: n // CHUNK]), sum(b["fails"][: n // CHUNK])
        pa, pb = fa / n, fb / n
        sig = (pa * (1 - pa) / n) ** 0.5
        print(json.dumps({"shots": n, "cpu_oracle_f64": {"fails": fa, "pL": pa}, "gpu": {"fails": fb, "pL": pb},
                          "sigma": sig, "abs_diff_in_sigma": abs(pa - pb) / sig}))
        return
    shots, out = int(sys.argv[2]), sys.argv[3]
    nch = shots // CHUNK
    t0 = time.time()
    if mode == "cpu":
        import multiprocessing as mp
        procs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
        with mp.Pool(procs) as pool:
            res = sorted(pool.map(cpu_chunk, range(nch), chunksize=1))
        rec = {"decoder": "oracle/qd_oracle.c, double precision, ldpc update order, min-sum flooding max_iter=50 + OSD-0",
               "fails": [r[1] for r in res], "bp_converged": [r[2] for r in res], "seconds": time.time() - t0, "procs": procs}
    else:
        import torch, helpers
        from quits_amd.decoder.device import BatchDecoder, DemSampler, GF2Matrix, WindowGraph, count_mismatch
        H, L, pri = helpers.dem_matrices(NAME)
        smp, g = DemSampler(H, L, pri), WindowGraph(H, pri)
        dec, Lm = BatchDecoder(g, max_iter=MAX_ITER, osd_method="osd_0", edge_messages=(mode == "gpu-edge")), GF2Matrix(L)
        fails, conv = [], []
        for c in range(nch):
            det, obs = smp.sample(CHUNK, seed=SEED, shot0=c * CHUNK)
            bits, status = dec.decode(det)
            pred = torch.zeros((CHUNK, L.shape[0]), dtype=torch.uint8, device="cuda")
            Lm.xor_apply(bits, pred, accumulate=False)
            fails.append(int(count_mismatch(pred, obs).item())); conv.append(int(((status >> 16) & 1).sum().item()))
        rec = {"decoder": "libquits_amd.so, float, %s min-sum flooding max_iter=50 + OSD-0"
                          % ("one message per edge (ldpc's update order)" if mode == "gpu-edge" else "compressed"), "fails": fails,
               "bp_converged": conv, "seconds": time.time() - t0}
    rec.update({"config": NAME, "chunk": CHUNK, "seed": SEED, "shots": nch * CHUNK})
    json.dump(rec, open(out, "w"))
    print(mode, "shots", nch * CHUNK, "fails", sum(rec["fails"]), "pL %.5f" % (sum(rec["fails"]) / (nch * CHUNK)), "%.0f s" % rec["seconds"])


if __name__ == "__main__":
    main()
