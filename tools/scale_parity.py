#!/usr/bin/env python3
"""Device vs oracle on a large sample of the headline window for the post-processors other than OSD-0 (whose 200 000-shot
golden lives in tests/golden/ler): every shot's correction, convergence flag and iteration count must agree.
usage (GPU box): tools/scale_parity.py [shots]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import multiprocessing as mp
import numpy as np

NAME = "bb144_custom_r12_p0.003"
_G = {}


def _work(arg):
    lo, hi, osd, order = arg
    import helpers, oracle as orc
    if "g" not in _G:
        H, L, pri = helpers.dem_matrices(NAME)
        g = orc.Graph(H, pri); g.device_grid(50)
        _G["g"] = g; _G["synd"] = np.load(os.environ["QD_SCALE_SYND"])
    prm = orc.make_params("minimum_sum", "parallel", 50, osd, order, 1.0, orc.FORM_LDPC_F64)
    ref, flags = _G["g"].decode_batch(_G["synd"][lo:hi], prm)
    return lo, np.packbits(ref, axis=1), flags


def main():
    shots = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    import torch, helpers
    from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph, unpack_bits
    H, L, pri = helpers.dem_matrices(NAME)
    det, obs = DemSampler(H, L, pri).sample(shots, seed=9)
    synd = det.cpu().numpy()
    path = "/tmp/qd_scale_synd.npy"; np.save(path, synd); os.environ["QD_SCALE_SYND"] = path
    wg = WindowGraph(H, pri)
    ncpu = len(os.sched_getaffinity(0))
    for osd, order in (("lsd_0", 0), ("osd_cs", 1), ("osd_e", 6)):
        dec = BatchDecoder(wg, max_iter=50, osd_method=osd, osd_order=order)
        bits, status = dec.decode(det)
        err = unpack_bits(bits, wg.n).cpu().numpy(); st = status.cpu().numpy()
        t0 = time.time()
        step = max(64, shots // (8 * ncpu))
        jobs = [(lo, min(lo + step, shots), osd, order) for lo in range(0, shots, step)]
        with mp.get_context("fork").Pool(ncpu) as pool:
            parts = pool.map(_work, jobs)
        ref = np.zeros_like(err); flags = np.zeros((shots, 4), np.int64)
        for lo, packed, fl in parts:
            k = fl.shape[0]
            ref[lo:lo + k] = np.unpackbits(packed, axis=1)[:, :wg.n]; flags[lo:lo + k] = fl
        same = (err == ref).all(axis=1)
        ok_conv = np.array_equal((st >> 16) & 1, flags[:, 0]); ok_it = np.array_equal(st & 0x3FFF, flags[:, 1])
        print("%-6s order %d: %d shots, %d post-processed, corrections identical on %d, convergence flags %s, iteration counts %s, inexact-flagged %d  (oracle %.0f s on %d CPUs)"
              % (osd, order, shots, int(((st >> 17) & 1).sum()), int(same.sum()), "identical" if ok_conv else "DIFFER", "identical" if ok_it else "DIFFER",
                 int(((st >> 15) & 1).sum()), time.time() - t0, ncpu))
        assert same.all() and ok_conv and ok_it


if __name__ == "__main__":
    main()
