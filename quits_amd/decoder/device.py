"""Thin Python objects over the C ABI (include/quits_amd.h): device-resident window graphs, batch decoders and
GF(2) matrices.  PyTorch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _lib
from .base import as_csr_int32


def _torch():
    import torch
    return torch


def _stream_ptr():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class WindowGraph:
    """Device copy of one window check matrix + priors (qd_graph).  Stands in for the sparse-matrix half of
    `BpOsdDecoder(pcm, channel_probs=...)` (reference sliding_window.py:149)."""

    def __init__(self, pcm, priors, device: Optional[int] = None):
        L = _lib.require_gpu()
        torch = _torch()
        self.device = torch.cuda.current_device() if device is None else int(device)
        rp, ci, (m, n) = as_csr_int32(pcm)
        pri = np.ascontiguousarray(np.broadcast_to(np.asarray(priors, dtype=np.float64), (n,)))
        self.m, self.n = int(m), int(n)
        self.words = (self.n + 31) // 32
        h = C.c_void_p()
        _lib.check(L.qd_graph_create(self.m, self.n, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                     pri.ctypes.data_as(C.c_void_p), self.device, C.byref(h)))
        self._h = h
        self._L = L
        self.priors = pri

    def info(self) -> dict:
        arr = (C.c_int32 * 12)()
        _lib.check(self._L.qd_graph_info_ex(self._h, arr, 12))
        keys = ("m", "n", "nnz", "max_row_weight", "max_col_weight", "bp_threads", "bp_lds_bytes", "osd_threads",
                "osd_lds_bytes", "rank", "scatter_walk_cycles", "scatter_walk_ideal")
        return dict(zip(keys, [int(x) for x in arr]))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self._L.qd_graph_destroy(self._h)
            except Exception:
                pass
            self._h = None


class BatchDecoder:
    """qd_decoder: BP(+OSD-0) over a batch of shots for one window."""

    def __init__(self, graph: WindowGraph, bp_method="minimum_sum", schedule="parallel", max_iter=0,
                 osd_method="osd_0", osd_order=0, ms_scaling_factor=1.0, edge_messages=False, raw_llr=False):
        """bp_method 'minimum_sum' + schedule 'parallel' runs in the compressed LDS kernel; every other pair -- and that one
        too when `edge_messages` is set -- in the one-message-per-edge kernel (csrc/bp_general.hip).  `raw_llr` keeps the
        channel LLRs off the binary grid (QD_FLAG_RAW_LLR: round-1 float arithmetic, validation only)."""
        self.graph = graph
        L = graph._L
        try:
            prm = _lib.QdParams(_lib.QD_BP[_norm(bp_method)], _lib.QD_SCHEDULE[_norm(schedule)], int(max_iter),
                                _lib.QD_OSD[_norm(osd_method)], int(osd_order), (1 if edge_messages else 0) | (2 if raw_llr else 0),
                                float(ms_scaling_factor))
        except KeyError as exc:
            raise ValueError("unknown decoder option %s" % exc) from exc
        h = C.c_void_p()
        rc = L.qd_decoder_create(graph._h, C.byref(prm), C.byref(h))
        if rc == -2:
            raise NotImplementedError(L.qd_last_error().decode())
        _lib.check(rc)
        self._h = h
        self._L = L
        self.params = prm

    def info(self) -> dict:
        """Arithmetic of this decoder (qd_decoder_info): LLR grid bits (-1 = float LLRs as they are), coarse grid, kernel."""
        arr = (C.c_int32 * 4)()
        _lib.check(self._L.qd_decoder_info(self._h, arr))
        post = {0: "none", 1: "qd_osd0_sr_kernel", 2: "qd_osd0_reg_kernel", 3: "qd_osd0_reg_kernel<row form>", 4: "qd_osdcs_kernel",
                5: "qd_lsd0_kernel"}.get(int(self._L.qd_decoder_postproc_kernel(self._h)), "?")
        return {"llr_grid_bits": int(arr[0]), "llr_coarse_bits": int(arr[1]), "edge_kernel": bool(arr[2]),
                "scatter_kernel": bool(arr[3]), "scatter_wide_kernel": int(arr[3]) == 2, "post_kernel": post}

    def set_workspace_limit(self, nbytes: int):
        """Cap the HBM message workspace of the one-message-per-edge BP kernel (product_sum / serial); no effect on the LDS kernel."""
        _lib.check(self._L.qd_decoder_set_workspace_limit(self._h, int(nbytes)))

    def release_workspace(self):
        """Give the device workspace back; the next decode sizes it again."""
        _lib.check(self._L.qd_decoder_release_workspace(self._h))

    def reserve(self, max_batch: int):
        _lib.check(self._L.qd_decoder_reserve(self._h, int(max_batch)))

    def decode(self, det, det_offset: int = 0, upd=None, err_bits=None, status=None, stage: int = 3, stream=None):
        """det: cuda uint8 [B, stride]; upd: cuda uint8 [B, rows] or None.
        stage 1 = BP only, 2 = OSD over the shots the preceding stage-1 call parked (same arguments), 3 = both.
        Returns (err_bits int32 [B, words], status int32 [B])."""
        torch = _torch()
        assert det.is_cuda and det.dtype == torch.uint8 and det.dim() == 2 and det.stride(1) == 1
        B = det.shape[0]
        g = self.graph
        if err_bits is None:
            err_bits = torch.empty((B, g.words), dtype=torch.int32, device=det.device)
        if status is None:
            status = torch.empty((B,), dtype=torch.int32, device=det.device)
        if upd is not None:
            assert upd.is_cuda and upd.dtype == torch.uint8 and upd.dim() == 2 and upd.stride(1) == 1
            up, us, ur = _ptr(upd), upd.stride(0), upd.shape[1]
        else:
            up, us, ur = C.c_void_p(0), 0, 0
        sp = _stream_ptr() if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self._L.qd_decode_stage(self._h, _ptr(det), det.stride(0), int(det_offset), up, us, ur, B,
                                           _ptr(err_bits), _ptr(status), int(stage), sp))
        return err_bits, status

    def post_head_start(self, stream, microseconds: int = -1):
        """Two-stream drivers: behind a stage-1 call on `stream`, hold that stream briefly if this batch's post-processing is heavy (qd_decoder_post_head_start)."""
        _lib.check(self._L.qd_decoder_post_head_start(self._h, int(microseconds), C.c_void_p(stream.cuda_stream)))

    def osd0(self, det, llr, det_offset: int = 0, upd=None):
        """OSD-0 alone on caller-supplied posteriors llr: cuda float32 [B, n] (fault order).  Returns (err_bits, status)."""
        torch = _torch()
        assert det.is_cuda and det.dtype == torch.uint8 and det.dim() == 2 and det.stride(1) == 1
        assert llr.is_cuda and llr.dtype == torch.float32 and llr.shape == (det.shape[0], self.graph.n) and llr.is_contiguous()
        B = det.shape[0]
        err_bits = torch.empty((B, self.graph.words), dtype=torch.int32, device=det.device)
        status = torch.empty((B,), dtype=torch.int32, device=det.device)
        if upd is not None:
            up, us, ur = _ptr(upd), upd.stride(0), upd.shape[1]
        else:
            up, us, ur = C.c_void_p(0), 0, 0
        _lib.check(self._L.qd_osd0_batch(self._h, _ptr(det), det.stride(0), int(det_offset), up, us, ur, B, _ptr(llr),
                                         _ptr(err_bits), _ptr(status), _stream_ptr()))
        return err_bits, status

    def failed_llr(self, b: int):
        torch = _torch()
        out = torch.empty((self.graph.n,), dtype=torch.float32, device="cuda")
        _lib.check(self._L.qd_decoder_failed_llr(self._h, int(b), _ptr(out), _stream_ptr()))
        return out

    def debug_counters(self):
        arr = (C.c_uint64 * 16)()
        _lib.check(self._L.qd_decoder_debug_counters(self._h, arr))
        return [int(x) for x in arr]

    def set_profiling(self, on: bool):
        _lib.check(self._L.qd_decoder_set_profiling(self._h, 1 if on else 0))

    def profile(self, reset: bool = True) -> dict:
        arr = (C.c_double * 4)()
        _lib.check(self._L.qd_decoder_profile(self._h, arr, 1 if reset else 0))
        return {"bp_ms": arr[0], "osd_ms": arr[1], "bp_launches": int(arr[2]), "osd_launches": int(arr[3])}

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self._L.qd_decoder_destroy(self._h)
            except Exception:
                pass
            self._h = None


class GF2Matrix:
    """Sparse GF(2) matrix on the device (qd_spmat), CSR."""

    def __init__(self, mat, device: Optional[int] = None):
        L = _lib.require_gpu()
        torch = _torch()
        self.device = torch.cuda.current_device() if device is None else int(device)
        rp, ci, (r, c) = as_csr_int32(mat)
        self.shape = (int(r), int(c))
        h = C.c_void_p()
        _lib.check(L.qd_spmat_create(self.shape[0], self.shape[1], rp.ctypes.data_as(C.c_void_p),
                                     ci.ctypes.data_as(C.c_void_p), self.device, C.byref(h)))
        self._h = h
        self._L = L

    def xor_apply(self, err_bits, out, accumulate: bool, stream=None):
        """out[b, :nrows] (^)= A @ e_b mod 2, e_b = packed bits err_bits[b]."""
        sp = _stream_ptr() if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self._L.qd_gf2_spmv_batch(self._h, _ptr(err_bits), err_bits.stride(0), err_bits.shape[0], _ptr(out),
                                             out.stride(0), 1 if accumulate else 0, sp))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self._L.qd_spmat_destroy(self._h)
            except Exception:
                pass
            self._h = None


def unpack_bits(bits, nbits: int):
    torch = _torch()
    L = _lib.load()
    out = torch.empty((bits.shape[0], nbits), dtype=torch.uint8, device=bits.device)
    _lib.check(L.qd_unpack_bits(_ptr(bits), bits.stride(0), int(nbits), bits.shape[0], _ptr(out), out.stride(0),
                                _stream_ptr()))
    return out


def count_mismatch(pred, obs) -> "object":
    """Device int64 scalar tensor: number of shots with pred != obs on any bit."""
    torch = _torch()
    L = _lib.load()
    assert pred.shape == obs.shape and pred.dtype == torch.uint8 and obs.dtype == torch.uint8
    pred = pred.contiguous()
    obs = obs.contiguous()
    cnt = torch.zeros((1,), dtype=torch.int64, device=pred.device)
    _lib.check(L.qd_count_mismatch(_ptr(pred), _ptr(obs), pred.shape[1], pred.shape[0], _ptr(cnt), _stream_ptr()))
    return cnt


class DemSampler:
    """Device-side detector-error-model sampler (stands in for stim's detector sampler, simulation.py:23-27)."""

    def __init__(self, check_matrix, observable_matrix, priors, device: Optional[int] = None):
        from scipy.sparse import csr_matrix
        self.Ht = GF2Matrix(csr_matrix(check_matrix).T.tocsr(), device)
        self.Lt = GF2Matrix(csr_matrix(observable_matrix).T.tocsr(), device)
        self.m = self.Ht.shape[1]
        self.nobs = self.Lt.shape[1]
        self.priors = np.ascontiguousarray(priors, dtype=np.float64)
        assert self.priors.shape[0] == self.Ht.shape[0]

    def sample(self, shots: int, seed: int, shot0: int = 0):
        torch = _torch()
        det = torch.empty((shots, self.m), dtype=torch.uint8, device="cuda")
        obs = torch.empty((shots, self.nobs), dtype=torch.uint8, device="cuda")
        L = self.Ht._L
        _lib.check(L.qd_sample_dem(self.Ht._h, self.Lt._h, self.priors.ctypes.data_as(C.c_void_p),
                                   C.c_uint64(int(seed) & (2 ** 64 - 1)), int(shot0), int(shots), _ptr(det),
                                   det.stride(0), _ptr(obs), obs.stride(0), _stream_ptr()))
        return det, obs


def _norm(x):
    return x.lower() if isinstance(x, str) else x
