"""The path end to end against the reference's PUBLISHED results (executed notebook cells: the only outputs of the real
ldpc + Stim pipeline available here; tools/published_anchor.py lists them with their cells).  Each case repeats the cell's call
on >= 10^5 DEM-sampled shots; the failure probability measured on the device must lie inside the exact (Clopper-Pearson) 95 %
interval of the published k failures in n trials.  Statistical by nature (the notebooks ran 100-1000 Stim shots), but it is
the one check that involves ldpc's and Stim's real behaviour: the DEM priors, product-sum / serial BP, OSD-CS order 1, BP-LSD
order 1 and both sliding-window drivers with the wrapper's defaults."""
import os
import sys

import numpy as np
import pytest

import helpers
import oracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import published_anchor as pa  # noqa: E402

SHOTS = 1 << 17


def test_clopper_pearson_known_values():
    lo, hi = pa.clopper_pearson(0, 200)
    assert lo == 0.0 and abs(hi - (1 - 0.025 ** (1 / 200))) < 1e-12
    lo, hi = pa.clopper_pearson(24, 200)
    assert 0.078 < lo < 0.079 and 0.173 < hi < 0.174
    assert pa.clopper_pearson(100, 100) == (pytest.approx(0.025 ** (1 / 100)), 1.0)


def test_anchor_fixtures_are_the_notebooks_circuits():
    """Shapes printed by the notebooks: detection_events (100, 1836) for HGP R = 15 (04 cell 6); code sizes (06A cell 3)."""
    from quits_amd.dem import Circuit
    from quits_amd.decoder.base import detector_error_model_to_matrix
    H, L, pri = detector_error_model_to_matrix(Circuit(helpers.circuit_text("hgp225_cardinal_r15_p0.001")).detector_error_model())
    assert H.shape[0] == 1836 and L.shape[0] == 9
    cd = helpers.code("hgp225")
    assert cd["hz"].shape == (108, 225) and cd["lz"].shape == (9, 225)
    assert helpers.code("bb90")["lz"].shape[0] == 8                      # 'rank_lz': 8 (06B cell 3)
    for case in pa.CASES:
        assert "%.10f" % case[5] in pa.circuit_for(case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", pa.CASES, ids=[c[0] for c in pa.CASES])
def test_device_failure_rate_is_compatible_with_the_published_count(gpu, case):
    import torch
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.decoder.device import DemSampler
    from quits_amd.dem import Circuit
    cid, cell, _, _, _, p, _, _, kind, kw, k_pub, n_pub = case
    H, L, pri = detector_error_model_to_matrix(Circuit(pa.circuit_for(case)).detector_error_model())
    det, obs = DemSampler(H, L, pri).sample(SHOTS, seed=1)
    pred = pa.device_decode(case, det)
    pl = float((pred != obs.cpu().numpy()).any(axis=1).mean())
    lo, hi = pa.clopper_pearson(k_pub, n_pub)
    assert lo <= pl <= hi, "%s (%s): device pL %.5f outside the 95 %% interval [%.5f, %.5f] of the published %d / %d" % (
        cid, cell, pl, lo, hi, k_pub, n_pub)


@pytest.mark.gpu
def test_anchor_settings_device_equals_its_float_mirror(gpu):
    """The 06A cell's settings at p = 2e-3 (the informative point: 24 / 200 published), 96 shots: device == f32 mirror bit for
    bit, and the f64 oracle (ldpc's arithmetic) agrees on nearly every shot."""
    case = [c for c in pa.CASES if c[0] == "06A_hgp_p2e-3"][0]
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.dem import Circuit
    H, L, pri = detector_error_model_to_matrix(Circuit(pa.circuit_for(case)).detector_error_model())
    synd, obs, _ = orc.sample_dem(H, L, pri, seed=77, shot0=0, B=96)
    pred = pa.device_decode(case, synd)
    p32 = pa._oracle_worker((case, synd, orc.FORM_LDPC_F32))
    assert np.array_equal(pred, p32.astype(np.int64))
    p64 = pa._oracle_worker((case, synd, orc.FORM_LDPC_F64))
    assert (pred != p64).any(axis=1).sum() <= 12


@pytest.mark.gpu
@pytest.mark.parametrize("cid,B", [("06A_hgp_p2e-3", 4096), ("05_hgp_phenom_bplsd", 8192)])
def test_anchor_device_vs_double_precision_oracle_paired(gpu, cid, B):
    """the two informative anchor points -- 24 / 200 (BP-OSD, circuit level) and 20 / 100 (BP-LSD order 1,
    phenomenological) -- on common shots: the device (float product-sum) against the oracle in ldpc's arithmetic (double, exact
    LLRs), paired.  McNemar's |z| <= 3 on the discordant shots, and both failure rates inside the published interval.  (The
    circuit-level case costs 23 ms of oracle per shot and core: 4096 shots here, 8192 in profiles/r04_published_anchor.json,
    tools/published_anchor.py --oracle-shots 8192.)"""
    import multiprocessing as mp
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.dem import Circuit
    case = [c for c in pa.CASES if c[0] == cid][0]
    H, L, pri = detector_error_model_to_matrix(Circuit(pa.circuit_for(case)).detector_error_model())
    synd, obs, _ = orc.sample_dem(H, L, pri, seed=4242, shot0=0, B=B)
    pred = pa.device_decode(case, synd)
    nproc = max(1, min(16, len(os.sched_getaffinity(0))))
    orc.lib()
    with mp.get_context("fork").Pool(nproc) as pool:                  # children never touch the GPU
        p64 = pa.oracle_decode(case, synd, orc.FORM_LDPC_F64, pool, nproc)
    fd = (pred != obs).any(axis=1)
    fo = (p64 != obs).any(axis=1)
    only_dev, only_orc = int((fd & ~fo).sum()), int((fo & ~fd).sum())
    z = (only_dev - only_orc) / max(1.0, float(np.sqrt(only_dev + only_orc)))
    lo, hi = pa.clopper_pearson(case[10], case[11])
    assert abs(z) <= 3.0, (cid, only_dev, only_orc, z)
    assert lo <= fd.mean() <= hi and lo <= fo.mean() <= hi, (cid, float(fd.mean()), float(fo.mean()), lo, hi)
