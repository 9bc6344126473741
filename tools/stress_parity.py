#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (run on the GPU box): random sparse check matrices, priors and syndromes through
every decoder configuration; everything must match the CPU oracle bit for bit (hard decisions, convergence flags, iteration
counts, OSD use, pivot counts).  usage: tools/stress_parity.py [trials] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from scipy.sparse import csc_matrix
import oracle as orc
from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits


def random_graph(rng):
    m = int(rng.choice([rng.integers(4, 40), rng.integers(40, 300), rng.integers(300, 1200), rng.integers(1200, 2600)],
                       p=[0.3, 0.35, 0.3, 0.05]))
    n = int(m * rng.uniform(1.0, 5.0)) + 1
    maxw = int(rng.integers(1, 7))
    rows, cols = [], []
    for j in range(n):
        w = int(rng.integers(1, min(maxw, m) + 1))
        for r in rng.choice(m, size=w, replace=False):
            rows.append(int(r)); cols.append(j)
    # no empty row, row weight <= 255
    present = set(rows)
    for r in range(m):
        if r not in present:
            rows.append(r); cols.append(int(rng.integers(0, n)))
    H = csc_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(m, n))
    H.data[:] = 1
    H.sum_duplicates(); H.data[:] = 1
    if np.diff(H.tocsr().indptr).max() > 250:
        return None
    style = rng.integers(0, 3)
    if style == 0:
        pri = np.full(n, float(10 ** rng.uniform(-3, -0.8)))                      # equal priors: ties everywhere
    elif style == 1:
        pri = 10 ** rng.uniform(-3, -0.7, n)
    else:
        pri = rng.choice([0.001, 0.003, 0.01, 0.05], size=n)
    return H, pri


def run(trials=200, seed=1):
    rng = np.random.default_rng(seed)
    t0 = time.time(); ok = 0; skipped = 0; inexact = 0; inexact_same = 0
    configs = [("minimum_sum", "parallel", False), ("minimum_sum", "parallel", True), ("product_sum", "parallel", True),
               ("product_sum", "serial", True), ("minimum_sum", "serial", True)]
    for t in range(trials):
        g = random_graph(rng)
        if g is None:
            skipped += 1; continue
        H, pri = g
        m, n = H.shape
        B = int(rng.integers(1, 70))
        errs = (rng.random((B, n)) < pri[None, :] * rng.uniform(0.5, 3.0)).astype(np.uint8)
        synd = (errs @ H.T.toarray().astype(np.int64) % 2).astype(np.uint8)
        if rng.random() < 0.3:
            synd[rng.integers(0, B)] = rng.integers(0, 2, m)                       # arbitrary (possibly inconsistent) syndrome
        if rng.random() < 0.3:
            synd[rng.integers(0, B)] = 0
        method, sched, edge = configs[int(rng.integers(0, len(configs)))]
        osd, order = [("osd_0", 0), ("osd_off", 0), ("osd_cs", int(rng.integers(0, 6))), ("osd_e", int(rng.integers(0, 5))), ("osd_0", 0),
                      ("lsd_0", 0), ("osd_cs", int(rng.integers(1, 12))), ("lsd_cs", int(rng.integers(1, 9))), ("lsd_e", int(rng.integers(1, 7))),
                      ("lsd_cs", 1)][int(rng.integers(0, 10))]
        max_iter = int(rng.integers(1, 25)) if sched == "parallel" else int(rng.integers(1, 13))     # (serial: up to 12, so that the staged launches -- bounds 3, 6, 10 -- are drawn)
        alpha = float(rng.choice([1.0, 1.0, 0.0, 0.625]))
        try:
            wg = WindowGraph(H, pri)
            dec = BatchDecoder(wg, bp_method=method, schedule=sched, max_iter=max_iter, osd_method=osd, osd_order=order,
                               ms_scaling_factor=alpha, edge_messages=edge)
        except (NotImplementedError, Exception) as exc:          # windows the device path declares unsupported
            if "QD_E" in str(exc) or isinstance(exc, NotImplementedError) or "capacity" in str(exc).lower() or "exceed" in str(exc).lower() or "does not fit" in str(exc):
                skipped += 1; continue
            raise
        bits, status = dec.decode(torch.from_numpy(synd).cuda())
        err = unpack_bits(bits, n).cpu().numpy(); st = status.cpu().numpy()
        go, form = orc.device_arithmetic(H, pri, method, sched, max_iter, alpha)     # grid + double for flooding min-sum at alpha 1
        if edge and form == orc.FORM_COMPRESSED_F32:
            form = orc.FORM_LDPC_F32
        ref, flags, grid = go.decode_batch(synd, orc.make_params(method, sched, max_iter, osd, order, alpha, form), return_grid=True)
        exact = np.ones(B, bool)
        if not edge and go.grid[0] >= 0:
            assert np.array_equal((st >> 14) & 1, (grid[:, 0] != go.grid[0]).astype(int)), ("coarse grid", (t, m, n))
            # QD_STATUS_INEXACT: the exactness bound tripped on the coarse grid too (messages of weight-1 checks are +-FLT_MAX,
            # sums of them overflow).  The oracle's own bound must say the same; such shots are outside the bit-exact contract.
            assert np.array_equal((st >> 15) & 1, grid[:, 1]), ("inexact flag", (t, m, n))
            exact = grid[:, 1] == 0
            inexact += int((~exact).sum())
            inexact_same += int(((err == ref).all(axis=1) & ~exact).sum())
        elif edge and go.grid[0] >= 0 and (grid[:, 0] != go.grid[0]).any():
            skipped += 1; continue                   # the edge kernel has no coarse-grid pass: such a batch is not comparable
        tag = (t, m, n, B, method, sched, edge, osd, order, max_iter, alpha)
        assert np.array_equal(((st >> 16) & 1)[exact], flags[exact, 0]), ("converged", tag)
        nz = synd.any(axis=1) & exact
        assert np.array_equal((st & 0x3FFF)[nz], flags[nz, 1]), ("iterations", tag)
        assert np.array_equal(err[exact], ref[exact]), ("decisions", tag, np.nonzero((err != ref).any(axis=1) & exact)[0][:5])
        if osd != "osd_off":
            assert np.array_equal(((st >> 17) & 1)[exact], (1 - flags[:, 0])[exact]), ("osd flag", tag)
            if osd in ("osd_0", "lsd_0", "lsd_cs", "lsd_e") or order == 0:
                used = (((st >> 17) & 1) == 1) & exact
                assert np.array_equal(((st >> 20) & 0xFFF)[used], np.minimum(flags[used, 2], 4095)), ("pivots", tag)
                assert np.array_equal(((st >> 18) & 1)[used], (flags[used, 3] != 0).astype(int)), ("inconsistent", tag)
        ok += 1
    print("stress parity: %d cases identical, %d skipped (unsupported by the device path), %d shots flagged QD_STATUS_INEXACT by device and oracle alike (outside the contract; %d of them decoded identically anyway), %.0f s"
          % (ok, skipped, inexact, inexact_same, time.time() - t0))
    return ok, skipped


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
