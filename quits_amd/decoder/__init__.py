"""Decoder package: mirrors `quits.decoder` (/root/reference/src/quits/decoder/__init__.py:3-24)."""
from .base import (detector_error_model_to_matrix, dict_to_csc_matrix_column_row,
                   dict_to_csc_matrix_row_column, spacetime)
from .sliding_window import sliding_window_circuit_mem, sliding_window_phenom_mem
from .bposd import BpOsdDecoder, sliding_window_bposd_circuit_mem, sliding_window_bposd_phenom_mem
from .bplsd import BpLsdDecoder, sliding_window_bplsd_circuit_mem, sliding_window_bplsd_phenom_mem

__all__ = [
    "dict_to_csc_matrix_column_row",
    "dict_to_csc_matrix_row_column",
    "detector_error_model_to_matrix",
    "spacetime",
    "sliding_window_phenom_mem",
    "sliding_window_circuit_mem",
    "sliding_window_bposd_phenom_mem",
    "sliding_window_bposd_circuit_mem",
    "sliding_window_bplsd_phenom_mem",
    "sliding_window_bplsd_circuit_mem",
    "BpOsdDecoder",
    "BpLsdDecoder",
]
