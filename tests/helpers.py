"""Fixture loaders shared by the tests (data only; nothing here touches /root/reference)."""
import gzip
import json
import os

import numpy as np
from scipy.sparse import csc_matrix

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def circuit_text(name):
    with gzip.open(os.path.join(GOLD, "circuits", name + ".stim.gz"), "rb") as f:
        return f.read().decode()


def circuit_text_at_p(name, p_from, p_to):
    """The circuit the reference emits for ErrorModel(p_to, p_to, p_to, p_to), derived from the fixture made at p_from.
    With one rate on all four channels every noise argument is printed as the same '%.10f' literal (circuit.py:96-246),
    so the two texts differ by exactly that literal (tests/test_dem.py checks this on the six bb144 fixtures)."""
    old, new = "%.10f" % p_from, "%.10f" % p_to
    text = circuit_text(name)
    assert old in text
    return text.replace(old, new)


def circuit_index():
    return json.load(open(os.path.join(GOLD, "circuits", "index.json")))


def code(name):
    z = np.load(os.path.join(GOLD, "codes", name + ".npz"))
    out = {}
    for k in ("hz", "hx", "lz", "lx"):
        shp = tuple(int(x) for x in z[k + "_shape"])
        out[k] = np.unpackbits(z[k + "_bits"])[: shp[0] * shp[1]].reshape(shp)
    return out


def csc_from(z, prefix):
    idx = z[prefix + "_indices"]
    return csc_matrix((np.ones(len(idx), np.uint8), idx, z[prefix + "_indptr"]),
                      shape=tuple(int(x) for x in z[prefix + "_shape"]))


def windows_npz(name):
    return np.load(os.path.join(GOLD, "windows", name + ".npz"))


def dem_matrices(name):
    z = windows_npz(name)
    return csc_from(z, "H"), csc_from(z, "L"), z["priors"]


def window_set(name, W, F):
    """The reference's spacetime() output for (W, F) as a list of dicts, in the oracle's window format."""
    z = windows_npz(name)
    tag = "W%dF%d" % (W, F)
    nwin = int(z[tag + "_nwin"][0])
    hz_rows = None
    out = []
    for k in range(nwin):
        out.append({"H": csc_from(z, "%s_H%d" % (tag, k)), "L": csc_from(z, "%s_L%d" % (tag, k)),
                    "priors": z["%s_p%d" % (tag, k)],
                    "U": csc_from(z, "%s_U%d" % (tag, k)) if k < nwin - 1 else None})
    return out


def same_sparse(a, b):
    a = csc_matrix(a); a.sort_indices()
    b = csc_matrix(b); b.sort_indices()
    return a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)


# ---- the CPU oracle over several processes (the oracle is single-threaded C; the larger parity tests slice the shots) ----------
def _oracle_procs():
    return max(1, min(32, len(os.sched_getaffinity(0))))


def _sw_worker(args):
    import oracle as orc
    wins, nz, det, prm, device_grid = args
    return orc.sliding_window_decode(wins, nz, det, orc.make_params(*prm), device_grid=device_grid)


def oracle_sliding_window_parallel(wins, nz, det, prm, device_grid=False):
    """orc.sliding_window_decode over shot slices in a fork pool; prm = the argument tuple of orc.make_params.
    Returns (predictions, summed stats)."""
    import multiprocessing as mp
    n = _oracle_procs()
    parts = [p for p in np.array_split(np.ascontiguousarray(det), min(n, max(1, len(det) // 2))) if len(p)]
    with mp.get_context("fork").Pool(min(n, len(parts))) as pool:
        res = pool.map(_sw_worker, [(wins, nz, p, prm, device_grid) for p in parts])
    stats = {k: sum(r[1][k] for r in res) for k in res[0][1]}
    return np.concatenate([r[0] for r in res], axis=0), stats


def _batch_worker(args):
    import oracle as orc
    H, pri, synd, prm, max_iter_grid = args
    g = orc.Graph(H, pri)
    if max_iter_grid is not None:
        g.device_grid(max_iter_grid)
    return g.decode_batch(synd, orc.make_params(*prm))


def oracle_decode_batch_parallel(H, pri, synd, prm, device_grid_max_iter=None):
    """Graph.decode_batch over shot slices in a fork pool (device_grid_max_iter: put the LLRs on the device's grid for that max_iter)."""
    import multiprocessing as mp
    n = _oracle_procs()
    parts = [p for p in np.array_split(np.ascontiguousarray(synd), min(n * 4, max(1, len(synd) // 8))) if len(p)]
    with mp.get_context("fork").Pool(min(n, len(parts))) as pool:
        res = pool.map(_batch_worker, [(H, pri, p, prm, device_grid_max_iter) for p in parts])
    return np.concatenate([r[0] for r in res], axis=0), np.concatenate([r[1] for r in res], axis=0)
