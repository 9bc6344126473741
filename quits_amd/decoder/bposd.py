"""BP-OSD bindings of the sliding-window decoders, and the HIP-backed plug-in decoder class.

Mirrors `/root/reference/src/quits/decoder/bposd.py`:
  sliding_window_bposd_phenom_mem   <- :10-51
  sliding_window_bposd_circuit_mem  <- :54-86
where the reference plugs in `ldpc.bposd_decoder.BpOsdDecoder` (bposd.py:5), this module plugs in the
`BpOsdDecoder` below, which runs on the MI355X through libquits_amd.so.

Device path coverage (anything else raises NotImplementedError -- never a silent change of algorithm, and there is
no CPU fallback):  bp_method in {'product_sum', 'minimum_sum'}, schedule in {'parallel', 'serial'} (natural fault
order), osd_method in {'osd_0', 'osd_off', 'osd_cs' with osd_order <= 64, 'osd_e' with osd_order <= 15}: every
combination the reference wrapper can request, including its defaults ('product_sum', 'serial', 'osd_cs').
'minimum_sum' + 'parallel' is the fast pair (check state compressed into LDS, csrc/bp_kernels.hip); the other three run
one lane per shot with per-edge messages in HBM (csrc/bp_general.hip) -- correct, bandwidth-bound, several times slower.
Higher-order OSD compares candidates by integer costs round(log(1/p) * 2^18) (exact, order-independent sums); ldpc
sums doubles, so the two can only disagree on candidates whose costs tie to ~1e-5.
"""
from __future__ import annotations

import numpy as np

from .sliding_window import sliding_window_circuit_mem, sliding_window_phenom_mem


class BpOsdDecoder:
    """Plug-in decoder with the constructor/`decode` surface of `ldpc.bposd_decoder.BpOsdDecoder` as the reference
    uses it (constructed at sliding_window.py:61,69,149,152; `decode` called at :85,95,171,182).

    One shot per `decode` call means one tiny kernel launch plus two PCIe copies per call -- this class exists for
    drop-in compatibility and parity tests; throughput comes from `decode_batch` / the batched drivers.
    """

    def __init__(self, pcm, error_rate=None, error_channel=None, max_iter=0, bp_method="minimum_sum",
                 ms_scaling_factor=1.0, schedule="parallel", omp_thread_count=1, random_schedule_seed=0,
                 serial_schedule_order=None, osd_method="osd_0", osd_order=0, input_vector_type="syndrome",
                 channel_probs=None, **kwargs):
        from .device import BatchDecoder, WindowGraph
        if kwargs:
            raise TypeError("unexpected keyword argument(s): %s" % ", ".join(sorted(kwargs)))
        if channel_probs is not None:               # ldpc v1 spelling, the one the reference passes (sliding_window.py:148)
            error_channel = channel_probs
        if error_channel is None:
            if error_rate is None:
                raise ValueError("Please specify the error channel. Either: 1) error_rate: float or "
                                 "2) error_channel: list of floats of length equal to the block length of the code.")
            error_channel = float(error_rate)
        if str(input_vector_type).lower() not in ("syndrome", "auto"):
            raise NotImplementedError("only syndrome input is supported")
        if serial_schedule_order is not None or random_schedule_seed not in (0, None):
            raise NotImplementedError("the serial schedule runs in natural fault order only (ldpc's default)")
        self.graph = WindowGraph(pcm, error_channel)
        self._dec = BatchDecoder(self.graph, bp_method=bp_method, schedule=schedule, max_iter=max_iter,
                                 osd_method=osd_method, osd_order=osd_order, ms_scaling_factor=ms_scaling_factor)
        self.m, self.n = self.graph.m, self.graph.n
        self.converge = False
        self.iter = 0
        self.last_status = None

    def decode_batch(self, syndromes):
        """syndromes: [B, m] numpy/torch (any integer/bool dtype)  ->  numpy uint8 [B, n]."""
        import torch
        from .device import unpack_bits
        from .sliding_window import _to_device_samples
        if not isinstance(syndromes, torch.Tensor):
            syndromes = np.asarray(syndromes)
            if syndromes.ndim != 2 or syndromes.shape[1] != self.m:
                raise ValueError("syndromes must have shape [B, %d]" % self.m)
        det = _to_device_samples(syndromes)
        err_bits, status = self._dec.decode(det, 0, None)
        out = unpack_bits(err_bits, self.n)
        self.last_status = status.cpu().numpy()
        return out.cpu().numpy()

    def decode(self, syndrome):
        syndrome = np.asarray(syndrome)
        if syndrome.ndim != 1 or syndrome.shape[0] != self.m:
            raise ValueError("The syndrome must have length %d. Not %s." % (self.m, syndrome.shape))
        out = self.decode_batch(syndrome.reshape(1, -1))[0]
        st = int(self.last_status[0])
        self.converge = bool(st & (1 << 16))
        self.iter = st & 0x3FFF
        return out.astype(syndrome.dtype) if syndrome.dtype != np.bool_ else out


def sliding_window_bposd_phenom_mem(zcheck_samples, hz, lz, W, F, eff_error_rate_per_fault: float = None, max_iter=2,
                                    osd_order=0, bp_method='product_sum', schedule='serial', osd_method='osd_cs',
                                    tqdm_on=False, error_rate: float = None):
    """Phenomenological sliding-window BP-OSD (reference bposd.py:10-51); same signature and defaults.

    :return logical_z_pred: int64 (# trials, # logical qubits)
    """
    if eff_error_rate_per_fault is None:
        eff_error_rate_per_fault = error_rate          # deprecated alias, kept like the reference (bposd.py:33-34)
    if eff_error_rate_per_fault is None:
        raise ValueError("eff_error_rate_per_fault must be provided (or use deprecated error_rate).")
    opts = {'bp_method': bp_method, 'max_iter': max_iter, 'schedule': schedule, 'osd_method': osd_method,
            'osd_order': osd_order, 'error_rate': float(eff_error_rate_per_fault)}
    return sliding_window_phenom_mem(zcheck_samples, hz, lz, W, F, BpOsdDecoder, BpOsdDecoder, dict(opts), dict(opts),
                                     'decode', 'decode', tqdm_on=tqdm_on)


def sliding_window_bposd_circuit_mem(zcheck_samples, circuit, hz, lz, W, F, max_iter=2, osd_order=0,
                                     bp_method='product_sum', schedule='serial', osd_method='osd_cs', tqdm_on=False):
    """Circuit-level sliding-window BP-OSD on the space-time detector error model (reference bposd.py:54-86); same
    signature and defaults.  `circuit` may be a stim.Circuit, the circuit text, or a quits_amd.dem.Circuit.

    :return logical_z_pred: int64 (# trials, # logical qubits)
    """
    opts = {'bp_method': bp_method, 'max_iter': max_iter, 'schedule': schedule, 'osd_method': osd_method,
            'osd_order': osd_order}
    return sliding_window_circuit_mem(zcheck_samples, circuit, hz, lz, W, F, BpOsdDecoder, BpOsdDecoder, dict(opts),
                                      dict(opts), 'channel_probs', 'channel_probs', 'decode', 'decode', tqdm_on=tqdm_on)


__all__ = ["sliding_window_bposd_phenom_mem", "sliding_window_bposd_circuit_mem"]
