#!/usr/bin/env python3
"""Paired logical-error-rate study for the reference wrapper's OWN settings (bposd.py:54 defaults + the notebooks' max_iter = 10,
osd_order = 1: product_sum, serial, osd_cs) at the headline window (BB [[144,12,12]], R = 12, p = 0.003, single window):
the device's float product-sum path (csrc/bp_general.hip, clamped tanh, f32) against the oracle in ldpc's arithmetic (double,
libm tanh/log) on the SAME Philox-sampled syndromes.  (missing #3.)

  python tools/ler_productsum.py run <shots> <seed> <out.npz> [procs] [first shot]    # CPU oracle (f64), any machine
  python tools/ler_productsum.py gpu <oracle.npz> [...]  <out.json>                    # on the GPU box: device on the same shots + report
The oracle column costs ~0.1 core-seconds per shot; it is computed once, off the GPU box, and committed
(tests/golden/ler/bb144_ps_serial_osdcs1_*.npz: observable + predicted logical bits per shot as uint16, 4 bytes per shot)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

CHUNK, NAME = 500, "bb144_custom_r12_p0.003"
OPTS = dict(bp_method="product_sum", schedule="serial", max_iter=10, osd_method="osd_cs", osd_order=1)
_G = {}


def _setup():
    import helpers, oracle as orc
    from scipy.sparse import csr_matrix
    H, L, pri = helpers.dem_matrices(NAME)
    _G.update(H=H, L=L, pri=pri, Lc=csr_matrix(L, dtype=np.int32), orc=orc, g=orc.Graph(H, pri),
              w=(1 << np.arange(L.shape[0])).astype(np.int64))


def _chunk(args):
    c, seed, shot0 = args
    orc = _G["orc"]
    det, obs, _ = orc.sample_dem(_G["H"], _G["L"], _G["pri"], seed=seed, shot0=shot0 + c * CHUNK, B=CHUNK)
    prm = orc.make_params(OPTS["bp_method"], OPTS["schedule"], OPTS["max_iter"], OPTS["osd_method"], OPTS["osd_order"], 1.0, orc.FORM_LDPC_F64)
    err, flags = _G["g"].decode_batch(det, prm)
    pred = np.asarray((_G["Lc"] @ err.T.astype(np.int32)) % 2).T
    return c, (obs.astype(np.int64) @ _G["w"]).astype(np.uint16), (pred.astype(np.int64) @ _G["w"]).astype(np.uint16), flags[:, 0].astype(np.uint8)


def run(shots, seed, path, procs, shot0=0):
    import multiprocessing as mp
    nch = shots // CHUNK
    t0 = time.time()
    with mp.Pool(procs, initializer=_setup) as pool:
        res = sorted(pool.imap_unordered(_chunk, [(c, seed, shot0) for c in range(nch)], chunksize=1), key=lambda r: r[0])
    meta = dict(config=NAME, opts=OPTS, form="ldpc_f64 (double, libm, ldpc's update order, exact LLRs)", seed=seed, shot0=shot0,
                shots=nch * CHUNK, procs=procs, seconds=time.time() - t0)
    np.savez_compressed(path, obs=np.concatenate([r[1] for r in res]), pred=np.concatenate([r[2] for r in res]),
                        conv=np.packbits(np.concatenate([r[3] for r in res])), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print("wrote", path, "%.0f s" % meta["seconds"])


def device_predictions(seed, shot0, shots):
    """(obs, pred, converged) of the device path on shots [shot0, shot0 + shots) of `seed`, as uint16 logical words."""
    import torch, helpers
    from quits_amd.decoder.device import BatchDecoder, DemSampler, GF2Matrix, WindowGraph
    H, L, pri = helpers.dem_matrices(NAME)
    smp, g, Lm = DemSampler(H, L, pri), WindowGraph(H, pri), GF2Matrix(L)
    dec = BatchDecoder(g, **OPTS)
    w = (1 << torch.arange(L.shape[0], device="cuda")).to(torch.int32)
    B = 32768
    o, p, cv = [], [], []
    for c0 in range(0, shots, B):
        b = min(B, shots - c0)
        det, obs = smp.sample(b, seed=seed, shot0=shot0 + c0)
        bits, status = dec.decode(det)
        pred = torch.zeros((b, L.shape[0]), dtype=torch.uint8, device="cuda")
        Lm.xor_apply(bits, pred, accumulate=False)
        o.append((obs.to(torch.int32) * w).sum(1).to(torch.int16).cpu().numpy().view(np.uint16))
        p.append((pred.to(torch.int32) * w).sum(1).to(torch.int16).cpu().numpy().view(np.uint16))
        cv.append(((status >> 16) & 1).to(torch.uint8).cpu().numpy())
    return np.concatenate(o), np.concatenate(p), np.concatenate(cv)


def paired(fail_dev, fail_ref):
    from math import erfc, sqrt
    n = len(fail_ref)
    b, c = int((fail_dev & ~fail_ref).sum()), int((~fail_dev & fail_ref).sum())
    pr = float(fail_ref.mean())
    sigma = sqrt(max(pr * (1 - pr), 1e-12) / n)
    z = (b - c) / sqrt(b + c) if b + c else 0.0
    return dict(shots=n, failures_device=int(fail_dev.sum()), failures_oracle_f64=int(fail_ref.sum()), ler_device=float(fail_dev.mean()),
                ler_oracle_f64=pr, sigma=sigma, delta_in_sigma=float((fail_dev.mean() - pr) / sigma), discordant=[b, c], mcnemar_z=z,
                mcnemar_p=erfc(abs(z) / sqrt(2)))


def gpu(paths, out):
    rows, fd_all, fr_all = [], [], []
    for path in paths:
        z = np.load(path)
        meta = json.loads(bytes(z["meta"]).decode())
        t0 = time.time()
        obs, pred, conv = device_predictions(meta["seed"], meta["shot0"], meta["shots"])
        assert np.array_equal(obs, z["obs"]), "device sampler and oracle sampler disagree"
        fd, fr = pred != obs, z["pred"] != z["obs"]
        r = paired(fd, fr)
        r.update(file=os.path.basename(path), seed=meta["seed"], shot0=meta["shot0"], device_seconds=round(time.time() - t0, 1),
                 identical_predictions=float((pred == z["pred"]).mean()),
                 bp_converged_device=float(conv.mean()), bp_converged_oracle=float(np.unpackbits(z["conv"])[:len(conv)].mean()))
        rows.append(r); fd_all.append(fd); fr_all.append(fr)
        print(json.dumps(r), flush=True)
    pooled = paired(np.concatenate(fd_all), np.concatenate(fr_all))
    json.dump(dict(config=NAME, opts=OPTS, files=rows, pooled=pooled), open(out, "w"), indent=1)
    print("pooled", json.dumps(pooled))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else os.cpu_count(),
            int(sys.argv[6]) if len(sys.argv) > 6 else 0)
    elif sys.argv[1] == "gpu":
        gpu(sys.argv[2:-1], sys.argv[-1])
