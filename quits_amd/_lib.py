"""ctypes binding of libquits_amd.so (include/quits_amd.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible, construction of any decoder
raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QUITS_AMD_LIB", os.path.join(_HERE, "lib", "libquits_amd.so"))

QD_BP = {"product_sum": 0, "ps": 0, "prod_sum": 0, "minimum_sum": 1, "min_sum": 1, "ms": 1, 0: 0, 1: 1}
QD_SCHEDULE = {"parallel": 0, "p": 0, "serial": 1, "s": 1, 0: 0, 1: 1}
QD_OSD = {"osd_off": 0, "off": 0, "osd_0": 1, "osd0": 1, "osd_e": 2, "osde": 2, "exhaustive": 2,
          "osd_cs": 3, "osdcs": 3, "combination_sweep": 3, "lsd_0": 4, "lsd_e": 5, "lsd_cs": 6, 0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 6}

STATUS_ITER_MASK = 0x3FFF
STATUS_COARSE_GRID = 1 << 14
STATUS_INEXACT = 1 << 15
STATUS_CONVERGED = 1 << 16
STATUS_OSD = 1 << 17
STATUS_INCONSISTENT = 1 << 18
STATUS_ZERO = 1 << 19


class QdParams(C.Structure):
    _fields_ = [("bp_method", C.c_int32), ("schedule", C.c_int32), ("max_iter", C.c_int32),
                ("osd_method", C.c_int32), ("osd_order", C.c_int32), ("reserved", C.c_int32),
                ("ms_scaling_factor", C.c_double)]


class QdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libquits_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None

EXPORTS = [
    "qd_version", "qd_last_error", "qd_device_count", "qd_graph_create", "qd_graph_destroy", "qd_graph_info", "qd_graph_info_ex",
    "qd_decoder_create", "qd_decoder_info", "qd_decoder_postproc_kernel", "qd_decoder_destroy", "qd_decoder_reserve", "qd_decoder_set_workspace_limit", "qd_decoder_release_workspace", "qd_decode_batch", "qd_decode_stage", "qd_osd0_batch", "qd_decoder_failed_llr",
    "qd_decoder_set_profiling", "qd_decoder_profile", "qd_decoder_post_head_start", "qd_decoder_debug_counters", "qd_spmat_create", "qd_spmat_destroy", "qd_gf2_spmv_batch",
    "qd_unpack_bits", "qd_count_mismatch", "qd_sample_dem",
]


def load():
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "quits_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and this library links the system one by the same
    # SONAME.  Whichever is loaded first serves both; loading this library first and torch afterwards leaves two runtimes, and
    # the second one sees no device.  So torch (the owner of device memory and streams on this path) goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    L.qd_version.restype = C.c_int
    if L.qd_version() < 103:              # 103: qd_decoder_post_head_start (the pipelined driver); 102: qd_graph_info_ex and the 10-entry qd_graph_info
        raise RuntimeError("quits_amd: %s is version %d, this package needs >= 103 -- rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)"
                           % (LIB_PATH, L.qd_version()))
    L.qd_last_error.restype = C.c_char_p
    L.qd_device_count.restype = C.c_int
    L.qd_graph_create.argtypes = [i32, i32, vp, vp, vp, i32, C.POINTER(vp)]
    L.qd_graph_destroy.argtypes = [vp]
    L.qd_graph_destroy.restype = None
    L.qd_graph_info.argtypes = [vp, vp]
    L.qd_graph_info_ex.argtypes = [vp, vp, i32]
    L.qd_decoder_create.argtypes = [vp, C.POINTER(QdParams), C.POINTER(vp)]
    L.qd_decoder_info.argtypes = [vp, vp]
    L.qd_decoder_postproc_kernel.argtypes = [vp]
    L.qd_decoder_postproc_kernel.restype = C.c_int
    L.qd_decoder_destroy.argtypes = [vp]
    L.qd_decoder_destroy.restype = None
    L.qd_decoder_reserve.argtypes = [vp, i64]
    L.qd_decoder_set_workspace_limit.argtypes = [vp, i64]
    L.qd_decoder_release_workspace.argtypes = [vp]
    L.qd_decode_batch.argtypes = [vp, vp, i64, i64, vp, i64, i32, i64, vp, vp, vp]
    L.qd_decode_stage.argtypes = [vp, vp, i64, i64, vp, i64, i32, i64, vp, vp, i32, vp]
    L.qd_osd0_batch.argtypes = [vp, vp, i64, i64, vp, i64, i32, i64, vp, vp, vp, vp]
    L.qd_decoder_failed_llr.argtypes = [vp, i64, vp, vp]
    L.qd_decoder_set_profiling.argtypes = [vp, i32]
    L.qd_decoder_post_head_start.argtypes = [vp, i32, vp]
    L.qd_decoder_profile.argtypes = [vp, vp, i32]
    L.qd_decoder_debug_counters.argtypes = [vp, vp]
    L.qd_spmat_create.argtypes = [i32, i32, vp, vp, i32, C.POINTER(vp)]
    L.qd_spmat_destroy.argtypes = [vp]
    L.qd_spmat_destroy.restype = None
    L.qd_gf2_spmv_batch.argtypes = [vp, vp, i64, i64, vp, i64, i32, vp]
    L.qd_unpack_bits.argtypes = [vp, i64, i32, i64, vp, i64, vp]
    L.qd_count_mismatch.argtypes = [vp, vp, i32, i64, vp, vp]
    L.qd_sample_dem.argtypes = [vp, vp, vp, u64, i64, i64, vp, i64, vp, i64, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise QdError(rc, load().qd_last_error().decode(errors="replace"))


def require_gpu():
    """Fail loudly when the device path cannot run (no silent CPU fallback)."""
    L = load()
    import torch
    if not torch.cuda.is_available() or L.qd_device_count() < 1:
        raise RuntimeError("quits_amd: no HIP device visible; the decoder has no CPU fallback "
                           "(the CPU restatement lives in oracle/ and is test infrastructure only)")
    return L
