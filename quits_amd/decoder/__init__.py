"""Decoder package: mirrors `quits.decoder` (/root/reference/src/quits/decoder/__init__.py:3-24)."""
from .base import (detector_error_model_to_matrix, dict_to_csc_matrix_column_row,
                   dict_to_csc_matrix_row_column, spacetime)
