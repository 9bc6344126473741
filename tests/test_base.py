"""Window slicing (quits_amd/decoder/base.py) against the reference's own outputs (golden G3, G7)."""
import warnings

import numpy as np
import pytest

import helpers
from quits_amd.decoder.base import spacetime, window_count, window_support_report, detector_error_model_to_matrix
from quits_amd.dem import Circuit

CASES = [("bb72_custom_r6_p0.003", "bb72", 6), ("bb144_custom_r12_p0.003", "bb144", 12),
         ("hgp225_cardinal_r3_p0.01", "hgp225", 3),
         ("qlp1020_cardinal_r20_p0.003", "qlp1020", 20)]      # BASELINE configs[4]: 20 windows of 1350 x 18900 (W = 3, F = 1)


@pytest.mark.parametrize("name,code,R", CASES)
def test_spacetime_matches_reference(name, code, R):
    z = helpers.windows_npz(name)
    circ = Circuit(helpers.circuit_text(name))
    hz = helpers.code(code)["hz"]
    tags = sorted({k.split("_")[0] for k in z.files if k.startswith("W")})
    assert tags
    for tag in tags:
        W, F = int(tag[1:tag.index("F")]), int(tag[tag.index("F") + 1:])
        ncr, w_last, whole = window_count(R, W, F)
        a, b, c, d = spacetime(circ, hz, W, F, ncr)
        assert len(a) == int(z[tag + "_nwin"][0]) == ncr + 1 and len(d) == ncr
        for k in range(len(a)):
            assert helpers.same_sparse(a[k], helpers.csc_from(z, "%s_H%d" % (tag, k)))
            assert helpers.same_sparse(b[k], helpers.csc_from(z, "%s_L%d" % (tag, k)))
            assert np.array_equal(c[k], z["%s_p%d" % (tag, k)])
            if k < ncr:
                assert helpers.same_sparse(d[k], helpers.csc_from(z, "%s_U%d" % (tag, k)))
        H, _, _ = detector_error_model_to_matrix(circ)
        rep = window_support_report(H, hz.shape[0], W, F, ncr)
        assert rep["lost_before"] == 0 and rep["lost_after"] == 0     # nothing falls between windows
        assert a[-1].shape[0] == w_last * hz.shape[0]


def test_window_table_matches_reference():
    """G7: (num_cor_rounds, W_last, warning) for an (R, W, F) grid, taken from the reference loop itself."""
    tab = np.load(helpers.GOLD + "/windows/window_table.npy")
    assert tab.shape[0] > 200
    for R, W, F, w_mid, w_last, warned in tab:
        ncr, wl, whole = window_count(int(R), int(W), int(F))
        assert wl == w_last and bool(warned) == whole
        assert (int(R) + 2 - wl) == F * ncr


def test_spacetime_errors():
    circ = Circuit(helpers.circuit_text("bb72_custom_r6_p0.003"))
    hz = helpers.code("bb72")["hz"]
    with pytest.raises(ValueError, match="F cannot be zero"):
        spacetime(circ, hz, 3, 0, 1)
    with pytest.raises(ValueError):                      # more windows than the record holds -> empty window
        spacetime(circ, hz, 3, 1, 12)
    # accepts circuit text and DEM objects alike
    a1 = spacetime(str(circ), hz, 3, 1, 6)[0]
    a2 = spacetime(circ.detector_error_model(), hz, 3, 1, 6)[0]
    assert all(helpers.same_sparse(x, y) for x, y in zip(a1, a2))
