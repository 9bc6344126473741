"""Hooks for the REAL denominators of SURVEY.md section 8(d): `ldpc.BpOsdDecoder` as the CPU baseline and Stim's detector sampler as
the input source.  Neither wheel exists in the build container or on the GPU box (profiles/r02_probe_ldpc_stim.txt), so every entry
point here probes first and reports the reason when the import fails; bench.py, tools/pin_ldpc.py and tests/test_refhooks.py (which
injects stub modules) are the callers.  Nothing here is on the product path.

Reference anchors: the import the reference makes is `from ldpc.bposd_decoder import BpOsdDecoder` (decoder/bposd.py:5) and
`from ldpc.bplsd_decoder import BpLsdDecoder` (decoder/bplsd.py:5); the sampler call is
`circuit.compile_detector_sampler(seed=seed).sample(shots=N, separate_observables=True)` (simulation.py:23-27)."""
from __future__ import annotations

import time

import numpy as np


def probe_ldpc():
    """(BpOsdDecoder class, version) when `ldpc` imports, else (None, reason)."""
    try:
        from ldpc.bposd_decoder import BpOsdDecoder
        import ldpc
        return BpOsdDecoder, str(getattr(ldpc, "__version__", "unknown"))
    except Exception as exc:                                      # ImportError, or a wheel built for another interpreter
        return None, "%s: %s" % (type(exc).__name__, exc)


def probe_ldpc_lsd():
    try:
        from ldpc.bplsd_decoder import BpLsdDecoder
        return BpLsdDecoder, None
    except Exception as exc:
        return None, "%s: %s" % (type(exc).__name__, exc)


def probe_stim():
    """(stim module, version) when `stim` imports, else (None, reason)."""
    try:
        import stim
        return stim, str(getattr(stim, "__version__", "unknown"))
    except Exception as exc:
        return None, "%s: %s" % (type(exc).__name__, exc)


def stim_sample(circuit_text: str, shots: int, seed: int):
    """The reference's get_stim_mem_result (simulation.py:8-28) on the circuit TEXT the fixtures hold: uint8 [shots, detectors],
    uint8 [shots, observables]."""
    stim, why = probe_stim()
    if stim is None:
        raise RuntimeError("stim is not importable (%s)" % why)
    circ = stim.Circuit(circuit_text)
    sampler = circ.compile_detector_sampler(seed=seed) if seed >= 0 else circ.compile_detector_sampler()
    det, obs = sampler.sample(shots=int(shots), separate_observables=True)
    return np.ascontiguousarray(det, dtype=np.uint8), np.ascontiguousarray(obs, dtype=np.uint8)


# ---- ldpc through the restated per-shot loop (quits_amd.decoder.sliding_window_circuit_mem with a foreign plug-in class = the
# reference's loop, sliding_window.py:143-186), over shot slices in a fork pool: decoders are rebuilt per worker, as 8(d) says
_W = {}


def _ldpc_worker(lo_hi):
    from quits_amd.decoder import sliding_window_circuit_mem
    lo, hi = lo_hi
    cls = _W["cls"]
    t = time.perf_counter()
    pred = sliding_window_circuit_mem(_W["det"][lo:hi], _W["circ"], _W["hz"], _W["lz"], _W["W"], _W["F"], cls, cls,
                                      dict(_W["opts"]), dict(_W["opts"]), "channel_probs", "channel_probs", "decode", "decode")
    return lo, np.asarray(pred, dtype=np.uint8), time.perf_counter() - t


def ldpc_window_loop(det, circ, hz, lz, W, F, opts, ncpu=1, cls=None):
    """Decode `det` (uint8 / bool [N, detectors], host) with ldpc's BpOsdDecoder through the sliding-window loop.
    Returns (pred uint8 [N, k], seconds of the first slice on one core, seconds of the whole sample on `ncpu` processes)."""
    import multiprocessing as mp
    import warnings
    if cls is None:
        cls, why = probe_ldpc()
        if cls is None:
            raise RuntimeError("ldpc is not importable (%s)" % why)
    det = np.ascontiguousarray(det)
    N = det.shape[0]
    _W.update(det=det, circ=circ, hz=hz, lz=lz, W=int(W), F=int(F), opts=dict(opts), cls=cls)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                            # the whole-history warning of a single window
        n1 = max(1, N // max(1, ncpu))
        _, p1, t1 = _ldpc_worker((0, n1))
        if ncpu > 1:
            sl = [(i * N // ncpu, (i + 1) * N // ncpu) for i in range(ncpu)]
            t = time.perf_counter()
            with mp.get_context("fork").Pool(ncpu) as pool:
                parts = sorted(pool.map(_ldpc_worker, sl), key=lambda r: r[0])
            ta = time.perf_counter() - t
            pred = np.concatenate([p[1] for p in parts])
        else:
            pred, ta = p1, t1
    return pred, n1, t1, ta
