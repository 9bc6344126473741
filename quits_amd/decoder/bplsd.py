"""BP-LSD bindings: names kept from `/root/reference/src/quits/decoder/bplsd.py:10,54`.  Localized-statistics
decoding is outside the hot path this package accelerates: calling these raises."""


def sliding_window_bplsd_phenom_mem(*args, **kwargs):
    raise NotImplementedError("BP-LSD is not implemented by quits_amd; use the BP-OSD decoders")


def sliding_window_bplsd_circuit_mem(*args, **kwargs):
    raise NotImplementedError("BP-LSD is not implemented by quits_amd; use the BP-OSD decoders")


__all__ = ["sliding_window_bplsd_phenom_mem", "sliding_window_bplsd_circuit_mem"]
