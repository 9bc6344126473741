"""Circuit -> detector error model, without Stim.

Stands in for `circuit.detector_error_model(decompose_errors=False)` at the reference call site
`/root/reference/src/quits/decoder/base.py:151` (Stim is an external C++ wheel that is absent from
the build image).  Semantics restated from Stim's error analyser:

* backward pass keeping, per qubit, the set of detectors/observables an X (resp. Z) error inserted
  at that point would flip;
* X_ERROR/Z_ERROR(p): one mechanism;  DEPOLARIZE1(p): X, Y, Z components, each with the
  independent-equivalent probability  q = 1/2 - 1/2*sqrt(1 - 4p/3);  DEPOLARIZE2(p): 15 components
  with  q = 1/2 - 1/2*(1 - 16p/15)**(1/8);
* mechanisms with identical (detectors, observables) symptom combine  p <- p(1-q) + q(1-p);
  empty symptoms are dropped;
* errors are emitted sorted lexicographically by target list, detectors before observables.  That is
  Stim's order inside one flush block; SURVEY.md App. C shows `spacetime()` cuts identical windows
  for this order and for Stim's block-by-block order.

The objects returned duck-type exactly the attributes `detector_error_model_to_matrix`
(`decoder/base.py:101-125`) touches, so the *reference* function can consume them unchanged (used to
generate tests/golden fixtures) and so can this package's restatement.
"""
from __future__ import annotations

import hashlib
import math
import os
import threading
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

from .stim_text import Op, flatten


class DemTarget:
    __slots__ = ("val", "_obs")

    def __init__(self, val: int, is_observable: bool):
        self.val = val
        self._obs = is_observable

    def is_relative_detector_id(self) -> bool:
        return not self._obs

    def is_logical_observable_id(self) -> bool:
        return self._obs

    def __repr__(self):
        return ("L%d" if self._obs else "D%d") % self.val


class DemInstruction:
    __slots__ = ("type", "_p", "_dets", "_obs")

    def __init__(self, p: float, dets: Tuple[int, ...], obs: Tuple[int, ...]):
        self.type = "error"
        self._p = p
        self._dets = dets
        self._obs = obs

    def args_copy(self) -> List[float]:
        return [self._p]

    def targets_copy(self) -> List[DemTarget]:
        return [DemTarget(d, False) for d in self._dets] + [DemTarget(o, True) for o in self._obs]

    def __repr__(self):
        t = " ".join(["D%d" % d for d in self._dets] + ["L%d" % o for o in self._obs])
        return "error(%.17g) %s" % (self._p, t)


class DetectorErrorModel:
    """Flat list of independent error mechanisms (already 'flattened': no repeat/shift)."""

    def __init__(self, errors: List[Tuple[float, Tuple[int, ...], Tuple[int, ...]]],
                 num_detectors: int, num_observables: int):
        self.errors = errors
        self.num_detectors = num_detectors
        self.num_observables = num_observables
        self.structure_key = None        # set by circuit_to_dem: DEMs of one circuit structure differ in their probabilities only

    def flattened(self):
        return [DemInstruction(p, d, o) for (p, d, o) in self.errors]

    @property
    def num_errors(self) -> int:
        return len(self.errors)

    def __str__(self):
        return "\n".join(repr(i) for i in self.flattened())


def _bits(x: int) -> List[int]:
    out = []
    while x:
        low = x & -x
        out.append(low.bit_length() - 1)
        x ^= low
    return out


_NOISE = ("X_ERROR", "Z_ERROR", "DEPOLARIZE1", "DEPOLARIZE2")

# Structure cache.  The reference's notebooks call the decoder once per physical error rate (doc/06B_end_to_end_demo_bb.ipynb cell 5)
# with circuits that differ in nothing but the noise arguments, and the analysis below is pure Python (11 s for the QLP [[1020,136]]
# circuit).  What depends on the arguments is only the probability attached to each symptom, so the symptoms, their order and -- per
# symptom -- the noise instructions that contribute, IN THE ORDER the pass meets them, are kept per circuit structure (everything but
# the noise arguments); another error rate then only replays  p <- p(1-q) + q(1-p)  in that order, vectorised by position, which gives
# the same floating-point numbers as the full pass, bit for bit.  QD_DEM_STRUCT_CACHE = structures kept (default 4, 0 = off).
_STRUCT_CACHE: "OrderedDict[str, dict]" = OrderedDict()
_STRUCT_STATS = {"hits": 0, "misses": 0}
_STRUCT_LOCK = threading.Lock()        # get / move_to_end / insert / evict / statistics under it; entries are never edited after insertion


def _struct_cap() -> int:
    try:
        return max(0, int(os.environ.get("QD_DEM_STRUCT_CACHE", "4")))
    except ValueError:
        return 4


def dem_struct_cache_info() -> dict:
    with _STRUCT_LOCK:
        return {"size": len(_STRUCT_CACHE), "capacity": _struct_cap(), **_STRUCT_STATS}


def dem_struct_cache_clear() -> None:
    with _STRUCT_LOCK:
        _STRUCT_CACHE.clear()
        _STRUCT_STATS["hits"] = 0
        _STRUCT_STATS["misses"] = 0


def _structure_key(ops, num_meas, num_det, num_obs) -> str:
    """Everything the analysis depends on except the VALUE of the noise arguments (whether one is zero does change the structure:
    a zero-probability mechanism is dropped)."""
    h = hashlib.sha1(("%d %d %d|" % (num_meas, num_det, num_obs)).encode())
    for op in ops:
        h.update(op.name.encode())
        if op.name in _NOISE:
            # keyed on the DERIVED probability: the full pass drops a mechanism whose probability is exactly 0.0, which for the
            # depolarising channels already happens at p ~ 1e-16 (the square / eighth root underflows), not only at p == 0
            h.update(b"+" if _mechanism_probability(op) > 0.0 else b"0")
        else:
            h.update(repr(op.arg).encode())
        h.update(np.asarray(op.targets, dtype=np.int64).tobytes())
        h.update(b";")
    return h.hexdigest()


def _mechanism_probability(op) -> float:
    """Independent-equivalent probability of ONE component of the noise instruction (see the module docstring)."""
    if op.name in ("X_ERROR", "Z_ERROR"):
        return op.arg
    if op.name == "DEPOLARIZE1":
        if op.arg > 0.75:
            raise ValueError("DEPOLARIZE1 probability above 3/4")
        return 0.5 - 0.5 * math.sqrt(1.0 - 4.0 * op.arg / 3.0)
    if op.arg > 15.0 / 16.0:
        raise ValueError("DEPOLARIZE2 probability above 15/16")
    return 0.5 - 0.5 * (1.0 - 16.0 * op.arg / 15.0) ** 0.125


def fold_steps(lists) -> list:
    """lists[i] = what is folded into entry i, in order  ->  steps[k] = (entries that have a k-th contribution, that contribution), as
    index arrays: replaying step 0, 1, 2, ... folds every entry in its own order."""
    import itertools
    n = len(lists)
    lens = np.fromiter((len(c) for c in lists), dtype=np.int64, count=n)
    total = int(lens.sum())
    if total == 0:
        return []
    owner = np.repeat(np.arange(n, dtype=np.int64), lens)
    start = np.cumsum(lens) - lens
    pos = np.arange(total, dtype=np.int64) - np.repeat(start, lens)
    what = np.fromiter(itertools.chain.from_iterable(lists), dtype=np.int64, count=total)
    order = np.argsort(pos, kind="stable")                     # (stable: entries stay in ascending order inside a step)
    cuts = np.cumsum(np.bincount(pos))
    steps, lo = [], 0
    for hi in cuts:
        sel = order[lo:hi]
        steps.append((owner[sel], what[sel]))
        lo = int(hi)
    return steps


def _replay_probabilities(st: dict, ops) -> np.ndarray:
    """Probabilities of the cached structure's symptoms for another set of noise arguments: the k-th contribution of every symptom
    is folded in at step k, i.e. in the order the full pass folds them."""
    q_of_op = np.zeros(len(ops), dtype=np.float64)
    for i in st["noise_ops"]:
        q_of_op[i] = _mechanism_probability(ops[i])
    prob = np.zeros(st["nsym"], dtype=np.float64)
    for k, (sym_idx, op_idx) in enumerate(st["steps"]):
        q = q_of_op[op_idx]
        if k == 0:
            prob[sym_idx] = q
        else:
            pk = prob[sym_idx]
            prob[sym_idx] = pk * (1.0 - q) + q * (1.0 - pk)
    return prob


def circuit_to_dem(text: str) -> DetectorErrorModel:
    """Backward Pauli-sensitivity analysis of a QUITS-dialect Stim circuit."""
    ops, num_meas, num_det, num_obs = flatten(text)
    cap = _struct_cap()
    skey = _structure_key(ops, num_meas, num_det, num_obs) if cap > 0 else None
    st = None
    if skey is not None:
        with _STRUCT_LOCK:
            st = _STRUCT_CACHE.get(skey)
            if st is not None:
                _STRUCT_CACHE.move_to_end(skey)
                _STRUCT_STATS["hits"] += 1
    if st is not None:
        prob = _replay_probabilities(st, ops)
        errors = [(float(prob[i]), d, o) for i, (d, o) in enumerate(st["rows"])]
        dem = DetectorErrorModel(errors, num_det, num_obs)
        dem.structure_key = skey
        return dem
    with _STRUCT_LOCK:
        _STRUCT_STATS["misses"] += 1
    # A symptom is the frozenset of the detectors (d) and observables (num_det + o) an error flips; XOR is the symmetric difference.
    # (Sets of a handful of integers hash and combine in ~100 ns; the 10^4-bit integer masks of rounds 1-3 cost a microsecond per
    # dictionary access on the QLP circuit: 10.4 -> 4-6 s there, the same output on every fixture; small circuits are unchanged.)
    empty = frozenset()
    meas_sens = [empty] * num_meas        # which detectors/observables include measurement k
    nq = 1 + max((max(op.targets) for op in ops if op.name not in ("DETECTOR", "OBSERVABLE_INCLUDE")
                  and op.targets), default=0)
    xs = [empty] * nq                      # flipped by an X error on q inserted *here*
    zs = [empty] * nq
    probs: Dict[frozenset, float] = {}
    contrib: Dict[frozenset, List[int]] = {}     # symptom -> indices of the noise instructions folded into it, in order
    cur_op = 0

    def add(sym, q: float):
        if not sym or q == 0.0:
            return
        p = probs.get(sym)
        if p is None:
            probs[sym] = q
            contrib[sym] = [cur_op]
        else:
            probs[sym] = p * (1.0 - q) + q * (1.0 - p)
            contrib[sym].append(cur_op)

    m = num_meas                           # running "measurements before this point" counter
    for cur_op in range(len(ops) - 1, -1, -1):
        op = ops[cur_op]
        name = op.name
        t = op.targets
        if name == "DETECTOR":
            bit = frozenset((int(op.arg),))
            for k in t:
                meas_sens[k] = meas_sens[k] ^ bit
        elif name == "OBSERVABLE_INCLUDE":
            bit = frozenset((num_det + int(op.arg),))
            for k in t:
                meas_sens[k] = meas_sens[k] ^ bit
        elif name == "CX":
            # forward: X_c -> X_c X_t ; Z_t -> Z_c Z_t  (pairs are applied in order; undo in reverse)
            for i in range(len(t) - 2, -1, -2):
                c, tg = t[i], t[i + 1]
                xs[c] = xs[c] ^ xs[tg]
                zs[tg] = zs[tg] ^ zs[c]
        elif name == "H":
            for q in t:
                xs[q], zs[q] = zs[q], xs[q]
        elif name == "M":
            for q in reversed(t):
                m -= 1
                xs[q] = xs[q] ^ meas_sens[m]
        elif name == "MX":
            for q in reversed(t):
                m -= 1
                zs[q] = zs[q] ^ meas_sens[m]
        elif name == "MR":
            for q in reversed(t):
                m -= 1
                xs[q] = meas_sens[m]       # reset erases later sensitivity, then the Z-measurement
                zs[q] = empty
        elif name in ("R", "RX"):
            for q in t:
                xs[q] = empty
                zs[q] = empty
        elif name == "X_ERROR":
            for q in t:
                add(xs[q], op.arg)
        elif name == "Z_ERROR":
            for q in t:
                add(zs[q], op.arg)
        elif name == "DEPOLARIZE1":
            q1 = _mechanism_probability(op)
            for q in t:
                x, z = xs[q], zs[q]
                add(x, q1)
                add(z, q1)
                add(x ^ z, q1)
        elif name == "DEPOLARIZE2":
            q2 = _mechanism_probability(op)
            for i in range(0, len(t), 2):
                a, b = t[i], t[i + 1]
                pa = (empty, xs[a], xs[a] ^ zs[a], zs[a])
                pb = (empty, xs[b], xs[b] ^ zs[b], zs[b])
                for ia in range(4):
                    for ib in range(4):
                        if ia or ib:
                            add(pa[ia] ^ pb[ib], q2)
        else:  # pragma: no cover - flatten() only emits the names above
            raise AssertionError(name)
    if m != 0:
        raise AssertionError("measurement bookkeeping is inconsistent")

    rows = []
    for sym, p in probs.items():
        dets = tuple(sorted(x for x in sym if x < num_det))
        obs = tuple(sorted(x - num_det for x in sym if x >= num_det))
        rows.append((dets, obs, p, sym))
    # Stim orders DemTargets with detectors before observables; compare target lists lexicographically
    rows.sort(key=lambda r: tuple(r[0]) + tuple(num_det + o for o in r[1]))
    errors = [(p, d, o) for (d, o, p, _) in rows]
    dem = DetectorErrorModel(errors, num_det, num_obs)
    if skey is not None:
        steps = fold_steps([contrib[r[3]] for r in rows])
        entry = {"rows": [(r[0], r[1]) for r in rows], "nsym": len(rows), "steps": steps,
                 "noise_ops": [i for i, op in enumerate(ops) if op.name in _NOISE and op.arg > 0.0]}
        with _STRUCT_LOCK:
            _STRUCT_CACHE[skey] = entry
            while len(_STRUCT_CACHE) > cap:
                _STRUCT_CACHE.popitem(last=False)
        dem.structure_key = skey
    return dem


class Circuit(str):
    """Circuit text with the one Stim method the decoder path calls (`decoder/base.py:151`)."""

    _dem_cache = None

    def detector_error_model(self, decompose_errors: bool = False, **_ignored) -> DetectorErrorModel:
        if decompose_errors:
            raise NotImplementedError("decompose_errors=True is not used by the QUITS decoder path")
        if self._dem_cache is None:
            self._dem_cache = circuit_to_dem(str(self))
        return self._dem_cache

    @property
    def num_detectors(self) -> int:
        return self.detector_error_model().num_detectors

    @property
    def num_observables(self) -> int:
        return self.detector_error_model().num_observables


def as_dem(circuit) -> DetectorErrorModel:
    """Accept a stim.Circuit (if Stim is installed), circuit text, a Circuit or a DEM-like object.

    A circuit is recognised by its `detector_error_model` method and asked for the DEM exactly as the reference does
    (`decoder/base.py:151`: `circuit.detector_error_model(decompose_errors=False)`); this has to be tested FIRST because a
    real stim.Circuit also carries `flattened()` and `num_detectors`.  Only an object without that method is taken for
    an already-built DEM (stim.DetectorErrorModel or the duck type `detector_error_model_to_matrix` reads)."""
    if isinstance(circuit, str):
        return (circuit if isinstance(circuit, Circuit) else Circuit(circuit)).detector_error_model()
    if hasattr(circuit, "detector_error_model"):
        return circuit.detector_error_model(decompose_errors=False)
    if hasattr(circuit, "flattened") and hasattr(circuit, "num_detectors"):
        return circuit
    raise TypeError("circuit must be a stim.Circuit, circuit text, quits_amd.dem.Circuit or a DEM")
