"""Detector-error-model -> matrices, and window slicing.

Host-side restatement of `/root/reference/src/quits/decoder/base.py`:
  detector_error_model_to_matrix  <- base.py:74-127
  spacetime                       <- base.py:134-190
Same names, argument meaning, return types (scipy.sparse.csc_matrix uint8 + float64 priors) and error
behaviour, so the reference's callers can switch imports.  Checked bit for bit against the reference functions
through tests/golden/{dem_merge,windows/*}.npz (tools/gen_fixtures.py G3/G4).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
from scipy.sparse import csc_matrix, csr_matrix

from collections import OrderedDict

from ..dem import as_dem

import threading as _threading

_MATRIX_LOCK = _threading.Lock()      # get / move_to_end / insert / evict under it; the copies handed out are made outside
_MATRIX_CACHE: "OrderedDict[str, tuple]" = OrderedDict()   # circuit structure (dem.structure_key) -> (check, observable, fold order of the priors)


def _csc_from_columns(cols: Sequence[Sequence[int]], nrows: int) -> csc_matrix:
    """Binary CSC matrix whose column j has ones at cols[j] (row order inside a column: ascending)."""
    counts = np.fromiter((len(c) for c in cols), dtype=np.int64, count=len(cols))
    indptr = np.zeros(len(cols) + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    indices = np.fromiter((r for c in cols for r in sorted(c)), dtype=np.int64, count=int(indptr[-1]))
    data = np.ones(indices.shape[0], dtype=np.uint8)
    return csc_matrix((data, indices, indptr), shape=(nrows, len(cols)))


def dict_to_csc_matrix_column_row(elements_dict, shape):
    """{column: rows} -> binary csc_matrix (reference base.py:26-47)."""
    ncols = shape[1]
    cols: List[Sequence[int]] = [() for _ in range(ncols)]
    for col, rows in elements_dict.items():
        cols[col] = tuple(rows)
    return _csc_from_columns(cols, shape[0])


def dict_to_csc_matrix_row_column(elements_dict, shape):
    """{rows (frozenset): column} -> binary csc_matrix (reference base.py:50-71)."""
    ncols = shape[1]
    cols: List[Sequence[int]] = [() for _ in range(ncols)]
    for rows, col in elements_dict.items():
        cols[col] = tuple(rows)
    return _csc_from_columns(cols, shape[0])


def detector_error_model_to_matrix(dem) -> Tuple[csc_matrix, csc_matrix, np.ndarray]:
    """DEM -> (check matrix [detectors x faults], observable matrix [observables x faults], priors).

    Behaviour kept from the reference (base.py:89-99), quirks included:
      * faults are keyed by their *detector set only*; a later mechanism with the same detectors is folded into
        the first one's probability  p <- p(1-q) + q(1-p)  and its observables are ignored (base.py:96);
      * column id = first-seen order in `dem.flattened()`;
      * an error without detectors is printed (base.py:114-115) and still becomes an (all-zero) column.
    """
    dem = as_dem(dem)
    skey = getattr(dem, "structure_key", None)
    hit = None
    if skey is not None:
        with _MATRIX_LOCK:
            hit = _MATRIX_CACHE.get(skey)
            if hit is not None:
                _MATRIX_CACHE.move_to_end(skey)
    if hit is not None:
        # same circuit structure, other probabilities (quits_amd/dem.py): the matrices are the cached ones, the priors are folded again in
        # the order the loop below folds them -- the same floating-point numbers
        check, obs, steps, no_det = hit
        for inst in no_det:                 # the reference prints an error without detectors on every call (base.py:114-115)
            print(inst)
        p_err = np.fromiter((e[0] for e in dem.errors), dtype=np.float64, count=len(dem.errors))
        priors = np.zeros(check.shape[1], dtype=np.float64)
        for k, (col_idx, err_idx) in enumerate(steps):
            p = p_err[err_idx]
            if k == 0:
                priors[col_idx] = p
            else:
                q = priors[col_idx]
                priors[col_idx] = q * (1 - p) + p * (1 - q)
        return check.copy(), obs.copy(), priors      # copies: this is a public function, callers may edit the result in place
    col_of: dict = {}
    det_sets: List[frozenset] = []
    obs_sets: List[frozenset] = []
    priors: List[float] = []
    members: List[List[int]] = []           # column -> the error instructions folded into it, in order
    n_err = 0
    no_det_insts: List[str] = []
    for inst in dem.flattened():
        kind = inst.type
        if kind == "error":
            p = inst.args_copy()[0]
            dets, obs = [], []
            for t in inst.targets_copy():
                if t.is_relative_detector_id():
                    dets.append(t.val)
                elif t.is_logical_observable_id():
                    obs.append(t.val)
            if not dets:
                print(inst)
                no_det_insts.append(str(inst))
            key = frozenset(dets)
            j = col_of.get(key)
            if j is None:
                col_of[key] = len(det_sets)
                det_sets.append(key)
                obs_sets.append(frozenset(obs))
                priors.append(p)
                members.append([n_err])
            else:
                q = priors[j]
                priors[j] = q * (1 - p) + p * (1 - q)
                members[j].append(n_err)
            n_err += 1
        elif kind in ("detector", "logical_observable"):
            continue
        else:
            raise NotImplementedError()
    check = _csc_from_columns(det_sets, dem.num_detectors)
    obs = _csc_from_columns(obs_sets, dem.num_observables)
    if skey is not None and n_err == len(getattr(dem, "errors", ())):
        from ..dem import fold_steps, _struct_cap
        cap = _struct_cap()                  # QD_DEM_STRUCT_CACHE sizes (and switches off) both structure caches
        if cap > 0:
            steps = fold_steps(members)
            entry = (check.copy(), obs.copy(), steps, tuple(no_det_insts))
            with _MATRIX_LOCK:
                _MATRIX_CACHE[skey] = entry
                while len(_MATRIX_CACHE) > cap:
                    _MATRIX_CACHE.popitem(last=False)
    return check, obs, np.array(priors)


def _last_nonempty_column(mat_csc: csc_matrix) -> int:
    """Index of the last column holding a one; ValueError if there is none (np.max of an empty set in the
    reference, base.py:163,169)."""
    nz = np.flatnonzero(np.diff(mat_csc.indptr) > 0)
    if nz.size == 0:
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    return int(nz[-1])


def window_count(num_rounds: int, W: int, F: int) -> Tuple[int, int, bool]:
    """(num_cor_rounds, W_last, whole_history) -- reference sliding_window.py:134-141 / :43-53."""
    if 2 + num_rounds - W >= 0:
        n = (2 + num_rounds - W) // F
        if (2 + num_rounds - W) % F != 0:
            n += 1
        whole = False
    else:
        n = 0
        whole = True
    return n, num_rounds + 2 - F * n, whole


def spacetime(circuit, hz, W, F, num_cor_rounds):
    """Cut the detector error matrix into sliding windows (reference base.py:134-190).

    Returns (window_check_set, window_observable_set, window_priors_set, window_update), lists of csc matrices /
    arrays with exactly the reference's shapes: window k covers detector rows [k*F*nz, (k*F+W)*nz) and the
    columns from `col_min` up to the last one touching those rows; the first `cor_max+1` of them (the last
    column touching the first F rounds) are the ones that get committed, `L_k`/`U_k` are restricted to them, and
    `U_k` is the single detector round right after the committed region.
    """
    if F == 0:
        raise ValueError("Input parameter F cannot be zero.")
    check, observable, priors = detector_error_model_to_matrix(as_dem(circuit))
    nz = hz.shape[0]
    checks, observables, prior_sets, updates = [], [], [], []
    col_min = 0
    for k in range(num_cor_rounds):
        win = check[k * F * nz:(k * F + W) * nz, col_min:]
        if win.shape[1] == 0:
            raise ValueError("There is no noise in one of the decoding window. This means there are redundant "
                             "detectors that do not check for any error.")
        col_max = _last_nonempty_column(win)
        win = win[:, :col_max + 1]
        checks.append(win)
        cor_max = _last_nonempty_column(win[:F * nz, :])
        observables.append(observable[:, col_min:col_min + cor_max + 1])
        prior_sets.append(priors[col_min:col_min + col_max + 1])
        updates.append(check[(k + 1) * F * nz:((k + 1) * F + 1) * nz, col_min:col_min + cor_max + 1])
        col_min += cor_max + 1
    checks.append(check[F * num_cor_rounds * nz:, col_min:])
    observables.append(observable[:, col_min:])
    prior_sets.append(priors[col_min:])
    return checks, observables, prior_sets, updates


def window_support_report(check: csc_matrix, nz: int, W: int, F: int, num_cor_rounds: int) -> dict:
    """The reference's slicing silently assumes time-ordered columns.  Report what it would
    lose: committed columns with support before the window or beyond the carried round, and columns never
    decoded."""
    H = csc_matrix(check)
    n = H.shape[1]
    first_row = np.full(n, -1)
    last_row = np.full(n, -1)
    for j in range(n):
        rows = H.indices[H.indptr[j]:H.indptr[j + 1]]
        if rows.size:
            first_row[j], last_row[j] = rows.min(), rows.max()
    lost_before = lost_after = 0
    col_min = 0
    for k in range(num_cor_rounds):
        win = H[k * F * nz:(k * F + W) * nz, col_min:]
        col_max = _last_nonempty_column(win)
        cor_max = _last_nonempty_column(win[:F * nz, :col_max + 1])
        sl = slice(col_min, col_min + cor_max + 1)
        has = first_row[sl] >= 0
        lost_before += int(np.sum(has & (first_row[sl] < k * F * nz)))
        lost_after += int(np.sum(has & (last_row[sl] >= ((k + 1) * F + 1) * nz)))
        col_min += cor_max + 1
    return {"lost_before": lost_before, "lost_after": lost_after, "last_window_cols": n - col_min}


def as_csr_int32(mat) -> Tuple[np.ndarray, np.ndarray, Tuple[int, int]]:
    A = csr_matrix(mat)
    A.sort_indices()
    A.sum_duplicates()
    if A.nnz and not np.all(A.data % 2 == 1):
        A.data = A.data % 2
        A.eliminate_zeros()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.shape


__all__ = [
    "dict_to_csc_matrix_column_row",
    "dict_to_csc_matrix_row_column",
    "detector_error_model_to_matrix",
    "spacetime",
]
