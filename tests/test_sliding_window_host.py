"""Host-side sliding-window logic against the reference's own loops (golden G5).

G5 was produced by running the REFERENCE functions sliding_window_circuit_mem / sliding_window_phenom_mem
(quits/decoder/sliding_window.py) with the CPU oracle as plug-in decoder.  Here the same plug-in runs inside
(a) this package's restated per-shot loops and (b) the oracle's C loop; both must reproduce G5 bit for bit."""
import warnings

import numpy as np
import pytest

import helpers
import oracle as orc
from quits_amd.decoder import (sliding_window_bposd_circuit_mem, sliding_window_bposd_phenom_mem,
                               sliding_window_circuit_mem, sliding_window_phenom_mem)
from quits_amd.decoder.base import spacetime, window_count
from quits_amd.decoder.sliding_window import phenom_window_matrices
from quits_amd.dem import Circuit

CASES = [("bb72_custom_r6_p0.003", "bb72", 6, ((3, 1, 20), (5, 3, 12), (8, 1, 30), (9, 2, 30)), 64),
         ("hgp225_cardinal_r3_p0.01", "hgp225", 3, ((3, 1, 15), (2, 1, 15)), 24),
         # X-basis memory experiment (basis="X": X-check detectors, lx observables) with four different channel rates
         ("bb72_custom_r2_xbasis_mixed", "bb72", 2, ((2, 1, 15), (4, 1, 15), (3, 2, 15)), 48)]


def _mats(code, name):
    """(check matrix, logicals) the decoder is handed: hx / lx for an X-basis circuit, hz / lz otherwise (the reference's
    argument names say `hz, lz` either way, decoder/bposd.py:54)."""
    cd = helpers.code(code)
    return (cd["hx"], cd["lx"]) if "xbasis" in name else (cd["hz"], cd["lz"])


def _golden(name):
    z = np.load(helpers.GOLD + "/loop/%s.npz" % name)
    shp = tuple(z["shape"])
    return z, np.unpackbits(z["syndromes"], axis=1)[:, :shp[1]]


@pytest.mark.parametrize("name,code,R,cases,nshots", CASES)
def test_host_loop_matches_reference_loop(name, code, R, cases, nshots):
    z, synd = _golden(name)
    hz, lz = _mats(code, name)
    cd = {"hz": hz, "lz": lz}
    circ = Circuit(helpers.circuit_text(name))
    for (W, F, mi) in cases:
        for grid, tag in ((None, "f64"), ("device", "grid")):
            opts = dict(bp_method="minimum_sum", max_iter=mi, schedule="parallel", osd_method="osd_0", osd_order=0,
                        form=orc.FORM_LDPC_F64, llr_grid=grid)
            d1, d2 = dict(opts), dict(opts)
            with warnings.catch_warnings(record=True) as wlog:
                warnings.simplefilter("always")
                pred = sliding_window_circuit_mem(synd[:nshots].astype(int), circ, cd["hz"], cd["lz"], W, F,
                                                  orc.OracleBpOsdDecoder, orc.OracleBpOsdDecoder, d1, d2,
                                                  "channel_probs", "channel_probs", "decode", "decode")
            assert pred.dtype == np.int64
            assert np.array_equal(pred, z["circ_W%dF%d_it%d_%s" % (W, F, mi, tag)][:nshots])
            assert (len(wlog) > 0) == (W > R + 2)                  # whole-history warning (sliding_window.py:138-140)
            assert "channel_probs" not in d1 and "channel_probs" not in d2     # caller's dicts are left alone


@pytest.mark.parametrize("name,code,R,cases,nshots", CASES)
def test_oracle_c_loop_matches_reference_loop(name, code, R, cases, nshots):
    z, synd = _golden(name)
    hz, lz = _mats(code, name)
    cd = {"hz": hz, "lz": lz}
    circ = Circuit(helpers.circuit_text(name))
    nz = cd["hz"].shape[0]
    for (W, F, mi) in cases:
        ncr, _, _ = window_count(R, W, F)
        checks, commits, priors, updates = spacetime(circ, cd["hz"], W, F, ncr)
        wins = [{"H": checks[k], "L": commits[k], "priors": priors[k], "U": updates[k] if k < ncr else None,
                 "row0": F * k * nz} for k in range(len(checks))]
        prm = orc.make_params("minimum_sum", "parallel", mi, "osd_0", 0, 1.0, orc.FORM_LDPC_F64)
        pred, stats = orc.sliding_window_decode(wins, nz, synd, prm, device_grid=True)
        assert np.array_equal(pred, z["circ_W%dF%d_it%d_grid" % (W, F, mi)])
        assert stats["bp_converged"] + stats["osd_calls"] == synd.shape[0] * len(wins)


@pytest.mark.parametrize("name,code,R,cases,nshots", CASES)
def test_phenom_host_loop_matches_reference_loop(name, code, R, cases, nshots):
    z, synd = _golden(name)
    hz, lz = _mats(code, name)
    cd = {"hz": hz, "lz": lz}
    for (W, F, mi) in cases[:2]:
        opts = dict(bp_method="minimum_sum", max_iter=mi, schedule="parallel", osd_method="osd_0", osd_order=0,
                    error_rate=0.03, form=orc.FORM_LDPC_F64, llr_grid="device")
        pred = sliding_window_phenom_mem(synd[:nshots].astype(int), cd["hz"], cd["lz"], W, F, orc.OracleBpOsdDecoder,
                                         orc.OracleBpOsdDecoder, dict(opts), dict(opts), "decode", "decode")
        assert np.array_equal(pred, z["phen_W%dF%d_it%d_grid" % (W, F, mi)][:nshots])


def test_phenom_window_matrices_shape():
    hz = helpers.code("bb72")["hz"]
    mid, last = phenom_window_matrices(hz, 3, 1, 4)
    nz, nq = hz.shape
    assert mid.shape == (3 * nz, 3 * (nq + nz)) and last.shape == (4 * nz, 4 * nq + 3 * nz)
    assert mid.nnz == 3 * hz.sum() + (3 + 2) * nz


def test_argument_errors():
    hz, lz = helpers.code("bb72")["hz"], helpers.code("bb72")["lz"]
    det = np.zeros((2, 36 * 8), dtype=int)
    with pytest.raises(ValueError, match="F cannot be zero"):
        sliding_window_phenom_mem(det, hz, lz, 3, 0, orc.OracleBpOsdDecoder, orc.OracleBpOsdDecoder, {}, {}, "decode", "decode")
    with pytest.raises(ValueError, match="eff_error_rate_per_fault"):
        sliding_window_bposd_phenom_mem(det, hz, lz, 3, 1)
    with pytest.raises(ValueError, match="F cannot be zero"):
        sliding_window_circuit_mem(det, Circuit(helpers.circuit_text("bb72_custom_r6_p0.003")), hz, lz, 3, 0,
                                   orc.OracleBpOsdDecoder, orc.OracleBpOsdDecoder, {}, {}, "channel_probs",
                                   "channel_probs", "decode", "decode")


def test_device_entry_points_fail_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device the BP-OSD entry points raise instead of decoding."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    hz, lz = helpers.code("bb72")["hz"], helpers.code("bb72")["lz"]
    det = np.zeros((2, 36 * 8), dtype=int)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        sliding_window_bposd_circuit_mem(det, Circuit(helpers.circuit_text("bb72_custom_r6_p0.003")), hz, lz, 3, 1,
                                         max_iter=5, bp_method="minimum_sum", schedule="parallel", osd_method="osd_0")
    with pytest.raises(RuntimeError):
        sliding_window_bposd_phenom_mem(det, hz, lz, 3, 1, eff_error_rate_per_fault=0.01, bp_method="minimum_sum",
                                        schedule="parallel", osd_method="osd_0")


def test_codecap_driver_matches_reference():
    """`get_codecap_pL` (simulation.py:31-61) with the oracle plug-in: same random stream, same logical error rate as the
    reference's own function run with the same plug-in (golden G8, tools/gen_fixtures.py codecap)."""
    import json, os, types
    from quits_amd.simulation import get_codecap_pL
    for ent in json.load(open(os.path.join(helpers.GOLD, "codecap.json"))):
        if ent["form"] != "f64" or ent["trials"] > 300:
            continue
        cd = helpers.code(ent["code"])
        cobj = types.SimpleNamespace(hz=cd["hz"], hx=cd["hx"], lz=cd["lz"], lx=cd["lx"])
        d = dict(ent["opts"], error_rate=ent["p"], form=orc.FORM_LDPC_F64)
        pl = get_codecap_pL(cobj, ent["p"], ent["trials"], orc.OracleBpOsdDecoder, d, basis=ent["basis"], seed=ent["seed"])
        assert pl == ent["pL"], ent
    with pytest.raises(ValueError):
        get_codecap_pL(cobj, 0.01, 1, orc.OracleBpOsdDecoder, d, basis="Y")
