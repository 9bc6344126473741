"""ctypes binding of oracle/liboracle.so (the CPU restatement; TEST INFRASTRUCTURE ONLY).

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The package `quits_amd`
never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BP_METHOD = {"product_sum": 0, "ps": 0, "prod_sum": 0, "minimum_sum": 1, "min_sum": 1, "ms": 1}
SCHEDULE = {"parallel": 0, "p": 0, "serial": 1, "s": 1}
OSD_METHOD = {"osd_off": 0, "off": 0, "osd_0": 1, "osd0": 1, "osd_e": 2, "osde": 2, "exhaustive": 2,
              "osd_cs": 3, "osdcs": 3, "combination_sweep": 3, "lsd_0": 4, "lsd0": 4, "lsd_e": 5, "lsde": 5, "lsd_cs": 6, "lsdcs": 6}
FORM_LDPC_F64, FORM_COMPRESSED_F32, FORM_COMPRESSED_F64, FORM_LDPC_F32 = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [("bp_method", C.c_int), ("schedule", C.c_int), ("max_iter", C.c_int),
                ("osd_method", C.c_int), ("osd_order", C.c_int), ("form", C.c_int),
                ("ms_scaling_factor", C.c_double)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("qd_oracle.c", "bp_core.inc", "oq_math.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


_NATIVE = {"want": False, "used": False, "flags": "-O3"}


def use_native(on: bool = True) -> None:
    """bench.py's cpu_baseline leg only (the CPU denominator is built `-O3 -march=native`): compile the same sources
    for THIS host's CPU into oracle/_native/ (never shipped: a -march=native object built in the build container could die with
    SIGILL on the GPU box's host) and load that.  Must be called before the first lib().  -ffp-contract=off stays, so the results
    are the portable build's bit for bit (tests/test_oracle.py::test_native_build_matches_portable)."""
    _NATIVE["want"] = bool(on)


def build_native() -> str:
    out_dir = os.path.join(_HERE, "_native")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liboracle_native.so")
    srcs = [os.path.join(_HERE, f) for f in ("qd_oracle.c", "bp_core.inc", "oq_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-fno-fast-math", "-ffp-contract=off", "-shared",
                               "-o", so, os.path.join(_HERE, "qd_oracle.c"), "-lm"])
    return so


def build_info() -> dict:
    return {"native": _NATIVE["used"], "flags": _NATIVE["flags"]}


def lib():
    global _LIB
    if _LIB is None:
        path = None
        if _NATIVE["want"]:
            try:
                path = build_native()
                _NATIVE["used"], _NATIVE["flags"] = True, "-O3 -march=native -ffp-contract=off"
            except (OSError, subprocess.CalledProcessError):
                path = None
        if path is None:
            path = build()
            _NATIVE["flags"] = "-O3 -ffp-contract=off (portable build)"
        L = C.CDLL(path)
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.oq_graph_create.restype = C.c_void_p
        L.oq_graph_create.argtypes = [C.c_int, C.c_int, i32p, i32p, f64p]
        L.oq_graph_destroy.argtypes = [C.c_void_p]
        L.oq_graph_quantize_llr.argtypes = [C.c_void_p, C.c_int]
        L.oq_graph_quantize_llr.restype = None
        L.oq_max_abs_llr.argtypes = [C.c_int]
        L.oq_max_abs_llr.restype = C.c_double
        L.oq_graph_set_coarse_grid.argtypes = [C.c_void_p, C.c_int]
        L.oq_graph_set_coarse_grid.restype = None
        L.oq_grid_bits_rule.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_int)]
        L.oq_bposd_decode_batch2.argtypes = [C.c_void_p, C.POINTER(Params), u8p, C.c_int64, u8p, i32p, i32p]
        L.oq_bp_decode.argtypes = [C.c_void_p, C.POINTER(Params), u8p, u8p, f64p, C.POINTER(C.c_int)]
        L.oq_osd_column_order.argtypes = [C.c_int, f64p, i32p]
        L.oq_gf2_rank.argtypes = [C.c_void_p]
        L.oq_osd0.argtypes = [C.c_void_p, u8p, f64p, C.c_int, u8p, i32p]
        L.oq_lsd0.argtypes = [C.c_void_p, u8p, f64p, u8p, i32p]
        L.oq_lsd.argtypes = [C.c_void_p, u8p, f64p, C.c_int, C.c_int, C.c_int, u8p, i32p]
        L.oq_osd_w.argtypes = [C.c_void_p, u8p, f64p, C.c_int, C.c_int, u8p]
        L.oq_osd_w_fixed.argtypes = [C.c_void_p, u8p, f64p, C.c_int, C.c_int, u8p, i32p]
        L.oq_fixed_weight.restype = C.c_uint32
        L.oq_fixed_weight.argtypes = [C.c_double]
        L.oq_bposd_decode_batch.argtypes = [C.c_void_p, C.POINTER(Params), u8p, C.c_int64, u8p, i32p]
        L.oq_csr_create.restype = C.c_void_p
        L.oq_csr_create.argtypes = [C.c_int, C.c_int, i32p, i32p]
        L.oq_csr_destroy.argtypes = [C.c_void_p]
        L.oq_sliding_window_decode.argtypes = [
            C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), i32p,
            C.c_int, C.c_int, C.c_int, C.POINTER(Params), u8p, C.c_int64, u8p,
            np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]
        L.oq_sample_dem.argtypes = [C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p, f64p, C.c_uint64,
                                    C.c_int64, C.c_int64, u8p, u8p, i32p]
        L.oq_philox.argtypes = [C.c_uint32] * 6 + [np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]
        L.oq_prob_threshold.restype = C.c_uint32
        L.oq_prob_threshold.argtypes = [C.c_double]
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.oq_math_f32.argtypes = [C.c_int, f32p, f32p, C.c_int64]
        L.oq_math_f32.restype = None
        L.oq_math_ucomb_f32.argtypes = [f32p, f32p, f32p, C.c_int64]
        L.oq_math_ucomb_f32.restype = None
        _LIB = L
    return _LIB


def make_params(bp_method="minimum_sum", schedule="parallel", max_iter=0, osd_method="osd_0", osd_order=0,
                ms_scaling_factor=1.0, form=FORM_LDPC_F64) -> Params:
    return Params(BP_METHOD[str(bp_method).lower()], SCHEDULE[str(schedule).lower()], int(max_iter),
                  OSD_METHOD[str(osd_method).lower()], int(osd_order), int(form), float(ms_scaling_factor))


def _csr(mat):
    from scipy.sparse import csr_matrix
    A = csr_matrix(mat)
    A.sort_indices()
    A.sum_duplicates()
    return A


class Graph:
    """One window's Tanner graph + priors."""

    def __init__(self, pcm, priors):
        A = _csr(pcm)
        self.m, self.n = A.shape
        pri = np.ascontiguousarray(np.broadcast_to(np.asarray(priors, dtype=np.float64), (self.n,)))
        self.priors = pri
        self.grid = (-1, -1)
        self._h = lib().oq_graph_create(self.m, self.n, A.indptr.astype(np.int32), A.indices.astype(np.int32), pri)
        if not self._h:
            raise ValueError("oq_graph_create failed (bad indices or column weight > 64)")

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.oq_graph_destroy(self._h)
            self._h = None

    def rank(self) -> int:
        return lib().oq_gf2_rank(self._h)

    def quantize_llr(self, frac_bits: int, coarse_bits: int = -1):
        """Round the channel LLRs to multiples of 2**-frac_bits (negative: back to the exact doubles).  With coarse_bits >= 0
        a shot whose exactness bound trips on the fine grid is decoded again on the coarse one, as the device does."""
        lib().oq_graph_quantize_llr(self._h, int(frac_bits))
        lib().oq_graph_set_coarse_grid(self._h, int(coarse_bits))
        self.grid = (int(frac_bits), int(coarse_bits))
        return self

    def device_grid(self, max_iter: int):
        """Put the LLRs on the grid libquits_amd.so picks for this graph and max_iter (flooding min-sum, ms_scaling 1)."""
        k, kc = grid_bits(self.priors, device_max_iter(max_iter, self.n))
        return self.quantize_llr(k, kc)

    def bp(self, syndrome, params: Params):
        s = np.ascontiguousarray(np.asarray(syndrome) % 2, dtype=np.uint8)
        dec = np.zeros(self.n, np.uint8)
        llr = np.zeros(self.n, np.float64)
        it = C.c_int(0)
        conv = lib().oq_bp_decode(self._h, C.byref(params), s, dec, llr, C.byref(it))
        if conv < 0:
            raise ValueError("unsupported bp_method/schedule for this arithmetic form")
        return bool(conv), dec, llr, it.value

    def osd0(self, syndrome, llr, stop_early=True):
        s = np.ascontiguousarray(np.asarray(syndrome) % 2, dtype=np.uint8)
        err = np.zeros(self.n, np.uint8)
        st = np.zeros(4, np.int32)
        lib().oq_osd0(self._h, s, np.ascontiguousarray(llr, dtype=np.float64), int(stop_early), err, st)
        return err, {"pivots": int(st[0]), "cols_examined": int(st[1]), "inconsistent": bool(st[2])}

    def lsd0(self, syndrome, llr):
        """BP-LSD's post-processing alone (LSD-0, one fault per growth step) on given soft information."""
        s = np.ascontiguousarray(np.asarray(syndrome) % 2, dtype=np.uint8)
        err = np.zeros(self.n, np.uint8)
        st = np.zeros(4, np.int32)
        lib().oq_lsd0(self._h, s, np.ascontiguousarray(llr, dtype=np.float64), err, st)
        return err, {"pivots": int(st[0]), "added": int(st[1]), "inconsistent": bool(st[2]), "rounds": int(st[3])}

    def lsd(self, syndrome, llr, lsd_method="lsd_cs", lsd_order=1, fixed=False):
        """BP-LSD's post-processing of any order on given soft information (lsd_order 0 = lsd0).  fixed=True: integer candidate
        costs (the HIP kernel's arithmetic)."""
        s = np.ascontiguousarray(np.asarray(syndrome) % 2, dtype=np.uint8)
        err = np.zeros(self.n, np.uint8)
        st = np.zeros(8, np.int32)
        lib().oq_lsd(self._h, s, np.ascontiguousarray(llr, dtype=np.float64), OSD_METHOD[lsd_method], int(lsd_order),
                     int(bool(fixed)), err, st)
        return err, {"pivots": int(st[0]), "added": int(st[1]), "inconsistent": bool(st[2]), "rounds": int(st[3]),
                     "grown": int(st[4]), "swept": int(st[5]), "replaced": int(st[6])}

    def osd_w(self, syndrome, llr, osd_method="osd_cs", osd_order=1, fixed=False):
        """OSD-CS / OSD-E.  fixed=True: integer candidate costs (the HIP kernel's arithmetic); returns (err, stats)."""
        s = np.ascontiguousarray(np.asarray(syndrome) % 2, dtype=np.uint8)
        err = np.zeros(self.n, np.uint8)
        if fixed:
            st = np.zeros(4, np.int32)
            lib().oq_osd_w_fixed(self._h, s, np.ascontiguousarray(llr, dtype=np.float64), OSD_METHOD[osd_method],
                                 int(osd_order), err, st)
            return err, {"pivots": int(st[0]), "winner": int(st[2]), "candidates": int(st[3])}
        lib().oq_osd_w(self._h, s, np.ascontiguousarray(llr, dtype=np.float64),
                       OSD_METHOD[osd_method], int(osd_order), err)
        return err

    def decode_batch(self, syndromes, params: Params, return_grid=False):
        S = np.ascontiguousarray(np.asarray(syndromes) % 2, dtype=np.uint8)
        S = S.reshape(-1, self.m)
        err = np.zeros((S.shape[0], self.n), np.uint8)
        flags = np.zeros((S.shape[0], 4), np.int32)
        grid = np.zeros((S.shape[0], 2), np.int32)
        rc = lib().oq_bposd_decode_batch2(self._h, C.byref(params), S, S.shape[0], err, flags, grid)
        if rc:
            raise ValueError("oracle decode failed (unsupported parameter combination)")
        return (err, flags, grid) if return_grid else (err, flags)


DEVICE_MAX_ITER = 16383      # libquits_amd.so keeps the iteration count in 14 status bits (QD_STATUS_ITER_MASK) and caps max_iter there


def device_max_iter(max_iter: int, n: int) -> int:
    """The iteration limit the device really runs for ldpc's `max_iter` (0 -> n) on a window of n faults."""
    return min(int(max_iter) if int(max_iter) > 0 else int(n), DEVICE_MAX_ITER)


def grid_bits(priors, max_iter: int):
    """(fine, coarse) LLR grid bits by the library's rule (oq_grid_bits_rule restates qd_decoder_create's)."""
    pri = np.asarray(priors, dtype=np.float64)
    mx = float(np.max(np.abs(np.log((1.0 - pri) / pri))))
    kc = C.c_int(0)
    k = lib().oq_grid_bits_rule(mx, int(max_iter), C.byref(kc))
    return int(k), int(kc.value)


def device_arithmetic(pcm, priors, bp_method="minimum_sum", schedule="parallel", max_iter=0, ms_scaling_factor=1.0):
    """(Graph, form) that reproduce libquits_amd.so bit for bit for these options: flooding min-sum with ms_scaling 1 runs
    exact arithmetic on the LLR grid -> ldpc's double-precision update order on the same grid; every other combination runs
    float arithmetic on float(log((1-p)/p)) -> the float mirrors."""
    g = Graph(pcm, priors)
    ms = BP_METHOD[str(bp_method).lower()] == 1
    par = SCHEDULE[str(schedule).lower()] == 0
    if ms and par and float(ms_scaling_factor) == 1.0:
        return g.device_grid(int(max_iter)), FORM_LDPC_F64
    return g, (FORM_COMPRESSED_F32 if (ms and par) else FORM_LDPC_F32)


def max_abs_llr(reset=False) -> float:
    return float(lib().oq_max_abs_llr(int(bool(reset))))


def column_order(llr):
    llr = np.ascontiguousarray(llr, dtype=np.float64)
    out = np.zeros(llr.shape[0], np.int32)
    lib().oq_osd_column_order(llr.shape[0], llr, out)
    return out


class OracleBpOsdDecoder:
    """Plug-in with ldpc.BpOsdDecoder's constructor/`decode` surface as the reference uses it
    (quits/decoder/sliding_window.py:61,149 construct; :85,171 call)."""

    def __init__(self, pcm, error_rate=None, error_channel=None, max_iter=0, bp_method="minimum_sum",
                 ms_scaling_factor=1.0, schedule="parallel", osd_method="osd_0", osd_order=0,
                 channel_probs=None, form=FORM_LDPC_F64, llr_grid=None, **_ignored):
        if channel_probs is not None:
            error_channel = channel_probs
        if error_channel is None:
            if error_rate is None:
                raise ValueError("error_rate or error_channel/channel_probs is required")
            error_channel = float(error_rate)
        self.graph = Graph(pcm, error_channel)
        if llr_grid == "device":       # the grid the HIP library uses for these options (flooding min-sum, ms_scaling 1)
            self.graph.device_grid(int(max_iter))
        elif llr_grid is not None:
            self.graph.quantize_llr(int(llr_grid))
        self.params = make_params(bp_method, schedule, max_iter, osd_method, osd_order, ms_scaling_factor, form)
        self.last_flags = None

    def decode(self, syndrome):
        err, flags = self.graph.decode_batch(np.asarray(syndrome).reshape(1, -1), self.params)
        self.last_flags = flags[0]
        return err[0]


def sliding_window_decode(windows, nz, samples, params: Params, device_grid=False):
    """windows: list of dicts {H (csr), priors, L (csr over committed cols), U (csr or None), row0}.
    device_grid: put every window's LLRs on the grid libquits_amd.so uses for it (flooding min-sum, ms_scaling 1)."""
    L = lib()
    samples = np.ascontiguousarray(np.asarray(samples) % 2, dtype=np.uint8)
    B, ndet = samples.shape
    graphs = [Graph(w["H"], w["priors"]) for w in windows]
    if device_grid:
        for g in graphs:
            g.device_grid(params.max_iter)
    nobs = windows[0]["L"].shape[0]
    keep = []

    def mk(mat, ncols):
        A = _csr(mat)
        h = L.oq_csr_create(A.shape[0], ncols, A.indptr.astype(np.int32), A.indices.astype(np.int32))
        keep.append(h)
        return h

    nw = len(windows)
    gs = (C.c_void_p * nw)(*[g._h for g in graphs])
    Ls = (C.c_void_p * nw)(*[mk(w["L"], w["L"].shape[1]) for w in windows])
    from scipy.sparse import csr_matrix
    Us = (C.c_void_p * nw)(*[mk(w["U"] if w.get("U") is not None else csr_matrix((nz, 1), dtype=np.uint8),
                                w["U"].shape[1] if w.get("U") is not None else 1) for w in windows])
    row0 = np.asarray([w["row0"] for w in windows], dtype=np.int32)
    pred = np.zeros((B, nobs), np.uint8)
    counters = np.zeros(4, np.int64)
    rc = L.oq_sliding_window_decode(nw, gs, Ls, Us, row0, nz, ndet, nobs, C.byref(params), samples, B, pred, counters)
    for h in keep:
        L.oq_csr_destroy(h)
    if rc:
        raise ValueError("oracle sliding-window decode failed")
    return pred, {"bp_converged": int(counters[0]), "osd_calls": int(counters[1]),
                  "bp_iters": int(counters[2]), "osd_inconsistent": int(counters[3])}


def sample_dem(H_csc, L_csc, priors, seed, shot0, B):
    from scipy.sparse import csc_matrix
    Hc = csc_matrix(H_csc); Hc.sort_indices()
    Lc = csc_matrix(L_csc); Lc.sort_indices()
    m, n = Hc.shape
    nobs = Lc.shape[0]
    synd = np.zeros((B, m), np.uint8)
    obs = np.zeros((B, nobs), np.uint8)
    nf = np.zeros(B, np.int32)
    lib().oq_sample_dem(m, n, nobs, Hc.indptr.astype(np.int32), Hc.indices.astype(np.int32),
                        Lc.indptr.astype(np.int32), Lc.indices.astype(np.int32),
                        np.ascontiguousarray(priors, dtype=np.float64), int(seed), int(shot0), int(B), synd, obs, nf)
    return synd, obs, nf


def math_f32(kind: str, x, x2=None):
    """The float functions of the f32 product-sum forms (oq_math.h): 'exp_neg' (+-e^-|x|, sign of x), 'neg_log' (-log(u)),
    'ucomb' ((a + b) / (1 + a b), two arguments)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    if kind == "ucomb":
        x2 = np.ascontiguousarray(x2, np.float32)
        assert x2.shape == x.shape
        lib().oq_math_ucomb_f32(x, x2, y, x.size)
    else:
        lib().oq_math_f32({"exp_neg": 0, "neg_log": 1}[kind], x, y, x.size)
    return y
