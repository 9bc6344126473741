"""BP-LSD bindings of the sliding-window decoders, and the HIP-backed plug-in class.

Mirrors `/root/reference/src/quits/decoder/bplsd.py`:
  sliding_window_bplsd_phenom_mem   <- :10-51
  sliding_window_bplsd_circuit_mem  <- :54-86
where the reference plugs in `ldpc.bplsd_decoder.BpLsdDecoder` (bplsd.py:5), this module plugs in the `BpLsdDecoder`
below: BP on the MI355X exactly as for BP-OSD (csrc/bp_kernels.hip / bp_general.hip), then localized statistics decoding
of the shots BP could not finish (csrc/lsd_kernels.hip: clusters grown from the unsatisfied checks, one fault per step in
order of posterior LLR, on-the-fly GF(2) elimination), one wavefront per shot.

Device path coverage: `lsd_method` in {'lsd_0', 'lsd_cs', 'lsd_e'} with `lsd_order` <= 64 ('lsd_cs') / <= 15 ('lsd_e') -- every
BP-LSD call the reference itself makes passes `lsd_order=1` (/root/reference/tests/test_decoders.py:136,
doc/05_decoder_variants.ipynb cell 9) -- and ldpc's default `bits_per_step = 1`.  A larger order or another step size raises
NotImplementedError -- never a silent change of algorithm; there is no CPU fallback.

This is LSD as published (Hillmann et al. 2024) with the open choices fixed as oracle/qd_oracle.c (oq_lsd) states them
(cluster order, tie breaks, which cluster absorbs, candidate costs log(1/p)): an LSD variant that is bit-exact against this
repo's oracle, not against ldpc's implementation (whose source is not available here); the anchor to ldpc is statistical
(tests/test_published_anchor.py: the BP-LSD failure rate published in doc/05_decoder_variants.ipynb).
"""
from __future__ import annotations

from .bposd import BpOsdDecoder
from .sliding_window import sliding_window_circuit_mem, sliding_window_phenom_mem

_LSD_METHODS = ("lsd_0", "lsd0", "lsd_e", "lsde", "lsd_cs", "lsdcs", 0, 1, 2)


def lsd_to_device_method(lsd_method, lsd_order, bits_per_step=1):
    """(osd_method, osd_order) of the device decoder for BpLsdDecoder's (lsd_method, lsd_order)."""
    key = lsd_method.lower() if isinstance(lsd_method, str) else lsd_method
    if key not in _LSD_METHODS:
        raise ValueError("lsd_method must be one of 'lsd_0', 'lsd_e', 'lsd_cs'")
    order = int(lsd_order)
    if order < 0:
        raise ValueError("lsd_order must be non-negative")
    if int(bits_per_step) != 1:
        raise NotImplementedError("BP-LSD on the device path grows clusters one fault per step (ldpc's default bits_per_step = 1)")
    if order == 0 or key in ("lsd_0", "lsd0", 0):          # with order 0 the three methods are the same decoder
        return "lsd_0", 0
    cs = key in ("lsd_cs", "lsdcs", 2)
    if order > (64 if cs else 15):
        raise NotImplementedError("BP-LSD on the device path implements lsd_order <= %d for %r; got lsd_order = %d" % (64 if cs else 15, lsd_method, order))
    return ("lsd_cs" if cs else "lsd_e"), order


class BpLsdDecoder(BpOsdDecoder):
    """Plug-in decoder with the constructor/`decode` surface of `ldpc.bplsd_decoder.BpLsdDecoder` as the reference uses it
    (kwargs built at bplsd.py:38-49,74-83; constructed and called by sliding_window.py:61,69,85,95,149,152,171,182)."""

    def __init__(self, pcm, error_rate=None, error_channel=None, max_iter=0, bp_method="minimum_sum",
                 ms_scaling_factor=1.0, schedule="parallel", omp_thread_count=1, random_schedule_seed=0,
                 serial_schedule_order=None, bits_per_step=1, lsd_order=0, lsd_method="lsd_0", input_vector_type="syndrome",
                 channel_probs=None, **kwargs):
        method, order = lsd_to_device_method(lsd_method, lsd_order, bits_per_step)
        super().__init__(pcm, error_rate=error_rate, error_channel=error_channel, max_iter=max_iter, bp_method=bp_method,
                         ms_scaling_factor=ms_scaling_factor, schedule=schedule, omp_thread_count=omp_thread_count,
                         random_schedule_seed=random_schedule_seed, serial_schedule_order=serial_schedule_order,
                         osd_method=method, osd_order=order,
                         input_vector_type=input_vector_type, channel_probs=channel_probs, **kwargs)


def sliding_window_bplsd_phenom_mem(zcheck_samples, hz, lz, W, F, eff_error_rate_per_fault: float = None, max_iter=2,
                                    lsd_order=0, bp_method='product_sum', schedule='serial', lsd_method='lsd_cs',
                                    tqdm_on=False, error_rate: float = None):
    """Phenomenological sliding-window BP-LSD (reference bplsd.py:10-51); same signature and defaults.

    :return logical_z_pred: int64 (# trials, # logical qubits)
    """
    if eff_error_rate_per_fault is None:
        eff_error_rate_per_fault = error_rate          # deprecated alias, kept like the reference (bplsd.py:33-34)
    if eff_error_rate_per_fault is None:
        raise ValueError("eff_error_rate_per_fault must be provided (or use deprecated error_rate).")
    opts = {'bp_method': bp_method, 'max_iter': max_iter, 'schedule': schedule, 'lsd_method': lsd_method,
            'lsd_order': lsd_order, 'error_rate': float(eff_error_rate_per_fault)}
    return sliding_window_phenom_mem(zcheck_samples, hz, lz, W, F, BpLsdDecoder, BpLsdDecoder, dict(opts), dict(opts),
                                     'decode', 'decode', tqdm_on=tqdm_on)


def sliding_window_bplsd_circuit_mem(zcheck_samples, circuit, hz, lz, W, F, max_iter=2, lsd_order=0,
                                     bp_method='product_sum', schedule='serial', lsd_method='lsd_cs', tqdm_on=False):
    """Circuit-level sliding-window BP-LSD on the space-time detector error model (reference bplsd.py:54-86); same
    signature and defaults.

    :return logical_z_pred: int64 (# trials, # logical qubits)
    """
    opts = {'bp_method': bp_method, 'max_iter': max_iter, 'schedule': schedule, 'lsd_method': lsd_method,
            'lsd_order': lsd_order}
    return sliding_window_circuit_mem(zcheck_samples, circuit, hz, lz, W, F, BpLsdDecoder, BpLsdDecoder, dict(opts),
                                      dict(opts), 'channel_probs', 'channel_probs', 'decode', 'decode', tqdm_on=tqdm_on)


__all__ = ["BpLsdDecoder", "sliding_window_bplsd_phenom_mem", "sliding_window_bplsd_circuit_mem"]
