"""Independent census of a circuit's detector error model by FORWARD propagation of every fault component, all components at once
(test infrastructure; the product's extractor quits_amd/dem.py works BACKWARDS from the detectors and shares nothing with this
file but the text parser).

Every component of every noise instruction (X / Z flips, the 3 Paulis of DEPOLARIZE1, the 15 of DEPOLARIZE2) is one bit position of
a packed Pauli frame  X[qubit, component words], Z[...]: a Clifford layer is a handful of row XORs / swaps over ALL components, a
measurement copies a row.  Out comes, per component, the set of detectors it flips -- and from the distinct non-empty sets the
quantities SURVEY.md Appendix B quotes for a DEM (columns, non-zeros, per-detector row weights), counted without the extractor."""
import numpy as np

from quits_amd.stim_text import flatten


def forward_detector_sets(text: str, lo: int = 0, hi: int = None, parsed=None):
    """-> (ndet, det_flip uint64 [ndet, words], ncomp, total): bit c - lo of det_flip[d] = component c flips detector d, for the
    components lo <= c < hi (all of them by default; large circuits are taken a range at a time)."""
    ops, nmeas, ndet, nobs = parsed if parsed is not None else flatten(text)
    nq = 1 + max(max(op.targets) for op in ops if op.name not in ("DETECTOR", "OBSERVABLE_INCLUDE") and op.targets)
    ncomp = 0
    for op in ops:
        t = len(op.targets)
        if op.name in ("X_ERROR", "Z_ERROR"):
            ncomp += t if op.arg > 0 else 0
        elif op.name == "DEPOLARIZE1":
            ncomp += 3 * t if op.arg > 0 else 0
        elif op.name == "DEPOLARIZE2":
            ncomp += 15 * (t // 2) if op.arg > 0 else 0
    total = ncomp
    hi = total if hi is None else min(hi, total)
    ncomp = hi - lo
    words = (ncomp + 63) // 64
    X = np.zeros((nq, words), np.uint64)
    Z = np.zeros((nq, words), np.uint64)
    meas = np.zeros((nmeas, words), np.uint64)
    one = np.uint64(1)

    def inject(F, q, c):
        if lo <= c < hi:
            F[q, (c - lo) >> 6] ^= one << np.uint64((c - lo) & 63)

    c = 0
    m = 0
    for op in ops:
        t = op.targets
        nm = op.name
        if nm == "CX":
            cs, ts = np.asarray(t[0::2]), np.asarray(t[1::2])
            X[ts] ^= X[cs]                       # (a layer's pairs are disjoint: fancy-index XOR is safe)
            Z[cs] ^= Z[ts]
        elif nm == "H":
            idx = np.asarray(t)
            tmp = X[idx].copy()
            X[idx] = Z[idx]
            Z[idx] = tmp
        elif nm in ("M", "MR", "MX"):
            idx = np.asarray(t)
            meas[m:m + len(t)] = Z[idx] if nm == "MX" else X[idx]
            m += len(t)
            if nm == "MR":
                X[idx] = 0
                Z[idx] = 0
        elif nm in ("R", "RX"):
            idx = np.asarray(t)
            X[idx] = 0
            Z[idx] = 0
        elif nm == "X_ERROR" and op.arg > 0:
            for q in t:
                inject(X, q, c); c += 1
        elif nm == "Z_ERROR" and op.arg > 0:
            for q in t:
                inject(Z, q, c); c += 1
        elif nm == "DEPOLARIZE1" and op.arg > 0:
            for q in t:
                inject(X, q, c); c += 1                          # X
                inject(X, q, c); inject(Z, q, c); c += 1          # Y
                inject(Z, q, c); c += 1                          # Z
        elif nm == "DEPOLARIZE2" and op.arg > 0:
            for j in range(0, len(t), 2):
                for pa in range(4):
                    for pb in range(4):
                        if pa == 0 and pb == 0:
                            continue
                        for q, p in ((t[j], pa), (t[j + 1], pb)):
                            if p in (1, 2):
                                inject(X, q, c)
                            if p in (2, 3):
                                inject(Z, q, c)
                        c += 1
    assert c == total and m == nmeas
    det = np.zeros((ndet, words), np.uint64)
    for op in ops:
        if op.name == "DETECTOR":
            d = int(op.arg)
            for k in op.targets:
                det[d] ^= meas[k]
    return ndet, det, ncomp, total


def census(text: str, chunk: int = 1 << 18):
    """Columns of the reference's check matrix (distinct non-empty detector sets, base.py:89-99 keys on the detector set alone),
    its non-zeros and its per-detector row weights, from the forward propagation alone."""
    parsed = flatten(text)
    ndet = parsed[2]
    rng = np.random.default_rng(12345)
    w1 = rng.integers(1, 2 ** 63 - 1, size=ndet, dtype=np.int64).astype(np.uint64)
    w2 = rng.integers(1, 2 ** 63 - 1, size=ndet, dtype=np.int64).astype(np.uint64)
    seen = set()
    row_w = np.zeros(ndet, np.int64)
    col_w = []
    lo, total = 0, None
    while total is None or lo < total:
        _, det, ncomp, total = forward_detector_sets(text, lo, lo + chunk, parsed)
        bits = np.unpackbits(det.view(np.uint8), axis=1, bitorder="little")[:, :ncomp]       # [ndet, ncomp]
        h1 = np.zeros(ncomp, np.uint64)
        h2 = np.zeros(ncomp, np.uint64)
        for d in range(ndet):                         # two independent 64-bit sums: a set's fingerprint
            sel = bits[d].astype(bool)
            h1[sel] += w1[d]
            h2[sel] += w2[d]
        nonempty = np.flatnonzero(bits.any(axis=0))
        key = np.stack([h1[nonempty], h2[nonempty]], axis=1)
        _, first = np.unique(key, axis=0, return_index=True)
        new = [i for i in first if (int(key[i, 0]), int(key[i, 1])) not in seen]
        seen.update((int(key[i, 0]), int(key[i, 1])) for i in new)
        cols = bits[:, nonempty[np.asarray(new, dtype=np.int64)]]            # one representative per distinct detector set not met before
        row_w += cols.sum(axis=1, dtype=np.int64)
        col_w.append(cols.sum(axis=0, dtype=np.int64))
        lo += chunk
    col_w = np.concatenate(col_w)
    return {"columns": int(len(col_w)), "nnz": int(col_w.sum()), "row_weights": row_w, "col_weights": col_w, "components": int(total)}
