"""GPU parity: libquits_amd.so (through the C ABI) against the CPU oracle on identical inputs.

Bar: bit-exact.  Flooding min-sum with ms_scaling 1 (the north star's pair, the only scaling the reference wrapper can
ask for) runs EXACT arithmetic on channel LLRs rounded to a binary grid (qd_decoder_info): it is compared with the oracle's
DOUBLE-precision form in ldpc's own update order (oracle/bp_core.inc bp_parallel_edge, REAL=double) on the same grid --
not with a mirror of the kernel.  Other options run float arithmetic and are compared with the float mirrors.  OSD is
integer work."""
import numpy as np
import pytest

import helpers
import oracle as orc

pytestmark = pytest.mark.gpu


def _oracle(H, pri, max_iter, osd="osd_0", alpha=1.0, order=0):
    """(graph, params) reproducing the device arithmetic for flooding min-sum (see module docstring)."""
    g, form = orc.device_arithmetic(H, pri, "minimum_sum", "parallel", max_iter, alpha)
    return g, orc.make_params("minimum_sum", "parallel", max_iter, osd, order, alpha, form)


def _gpu_decode(H, pri, synd, max_iter, osd="osd_0", alpha=1.0, order=0):
    import torch
    from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits
    g = WindowGraph(H, pri)
    d = BatchDecoder(g, max_iter=max_iter, osd_method=osd, osd_order=order, ms_scaling_factor=alpha)
    det = torch.from_numpy(np.ascontiguousarray(synd)).cuda()
    bits, status = d.decode(det)
    err = unpack_bits(bits, g.n).cpu().numpy()
    return err, status.cpu().numpy(), d


@pytest.mark.parametrize("name,shots", [("bb72_custom_r6_p0.003", 1500), ("hgp225_cardinal_r3_p0.01", 300)])
def test_sampler_matches_oracle(gpu, name, shots):
    from quits_amd.decoder.device import DemSampler
    H, L, pri = helpers.dem_matrices(name)
    det, obs = DemSampler(H, L, pri).sample(shots, seed=1234567, shot0=77)
    s_ref, o_ref, _ = orc.sample_dem(H, L, pri, seed=1234567, shot0=77, B=shots)
    assert np.array_equal(det.cpu().numpy(), s_ref)
    assert np.array_equal(obs.cpu().numpy(), o_ref)


@pytest.mark.parametrize("name,shots,max_iter,alpha", [
    ("bb72_custom_r6_p0.003", 1200, 30, 1.0),
    ("bb72_custom_r6_p0.003", 400, 25, 0.0),
    ("bb72_custom_r6_p0.003", 400, 25, 0.8125),
    ("hgp225_cardinal_r3_p0.01", 200, 20, 1.0),
    ("bb144_custom_r12_p0.003", 300, 50, 1.0),
])
def test_bp_bit_exact(gpu, name, shots, max_iter, alpha):
    H, L, pri = helpers.dem_matrices(name)
    synd, _, _ = orc.sample_dem(H, L, pri, seed=5, shot0=0, B=shots)
    synd[0] = 0                                         # all-zero syndrome short-circuit
    err, status, dec = _gpu_decode(H, pri, synd, max_iter, osd="osd_off", alpha=alpha)
    g, prm = _oracle(H, pri, max_iter, "osd_off", alpha)
    ref, flags, grid = g.decode_batch(synd, prm, return_grid=True)
    info = dec.info()
    if alpha == 1.0:
        assert (info["llr_grid_bits"], info["llr_coarse_bits"]) == g.grid == orc.grid_bits(pri, max_iter)
        assert np.array_equal((status >> 14) & 1, (grid[:, 0] != g.grid[0]).astype(int)), "coarse-grid flags differ"
        assert not (status & (1 << 15)).any()
    else:
        assert info["llr_grid_bits"] == -1
    conv = (status >> 16) & 1
    assert np.array_equal(conv, flags[:, 0]), "convergence flags differ"
    assert np.array_equal(status & 0x3FFF, flags[:, 1]), "iteration counts differ"
    assert np.array_equal(err, ref), "hard decisions differ"
    assert status[0] & (1 << 19) and not err[0].any()
    assert 0 < conv.mean() < 1 or shots < 50


def test_large_max_iter_keeps_a_fine_llr_grid(gpu):
    """ldpc's max_iter = 0 (-> n iterations): the fine grid stays at 2^-10 and only the shots whose exactness bound trips are
    decoded again on the rule's grid (the grid used to follow max_iter down to 2^-3 for every shot).  Device ==
    oracle bit for bit including which shots took the redo pass; and the predictions stay close to the exact-LLR double
    form (ldpc's arithmetic): LLRs within 5e-4 of ldpc's for every shot that stays within the fine grid's range."""
    name = "bb72_custom_r6_p0.003"
    H, L, pri = helpers.dem_matrices(name)
    shots = 400
    synd, obs, _ = orc.sample_dem(H, L, pri, seed=41, shot0=0, B=shots)
    err, status, dec = _gpu_decode(H, pri, synd, 0, osd="osd_0")
    info = dec.info()
    assert (info["llr_grid_bits"], info["llr_coarse_bits"]) == orc.grid_bits(pri, H.shape[1]) and info["llr_grid_bits"] == 10
    g, prm = _oracle(H, pri, 0, "osd_0")
    ref, flags, grid = g.decode_batch(synd, prm, return_grid=True)
    assert np.array_equal((status >> 14) & 1, (grid[:, 0] != g.grid[0]).astype(int)), "coarse-grid flags differ"
    assert 0 < ((status >> 14) & 1).sum() < shots // 2, "the redo pass should run for the few shots BP cannot finish, not for all"
    assert not (status & (1 << 15)).any()
    assert np.array_equal(status & 0x3FFF, flags[:, 1]) and np.array_equal(err, ref)
    exact, fl_exact = orc.Graph(H, pri).decode_batch(synd, orc.make_params("minimum_sum", "parallel", 0, "osd_0", 0, 1.0, orc.FORM_LDPC_F64))
    Ld = np.asarray(L.todense(), dtype=np.int64)
    p_dev, p_ex = (err.astype(np.int64) @ Ld.T) % 2, (exact.astype(np.int64) @ Ld.T) % 2
    f_dev, f_ex = (p_dev != obs).any(axis=1), (p_ex != obs).any(axis=1)
    # min-sum run to n iterations is chaotic on the shots that converge late: 2^-11 on the inputs moves a few per cent of the
    # predictions either way (22 of 400 here, 26 vs 34 failures) -- what must hold is that it is a few per cent, not a bias
    assert (p_dev != p_ex).any(axis=1).mean() < 0.1
    assert abs(int(f_dev.sum()) - int(f_ex.sum())) <= 3 * np.sqrt(max(f_ex.sum(), 1)) + 2, (f_dev.sum(), f_ex.sum())


@pytest.mark.parametrize("name,shots,max_iter", [
    ("bb72_custom_r6_p0.003", 1200, 20),
    ("hgp225_cardinal_r3_p0.01", 200, 10),
    ("bb144_custom_r12_p0.003", 400, 30),
])
def test_bposd_bit_exact(gpu, name, shots, max_iter):
    H, L, pri = helpers.dem_matrices(name)
    synd, _, _ = orc.sample_dem(H, L, pri, seed=11, shot0=0, B=shots)
    err, status, dec = _gpu_decode(H, pri, synd, max_iter, osd="osd_0")
    g, prm = _oracle(H, pri, max_iter, "osd_0")
    ref, flags = g.decode_batch(synd, prm)
    used_osd = (status >> 17) & 1
    assert np.array_equal(used_osd, 1 - flags[:, 0])
    assert used_osd.sum() > 10, "test does not exercise OSD"
    assert np.array_equal((status >> 20) & 0xFFF, np.minimum(flags[:, 2], 4095)), "pivot counts differ"
    bad = np.flatnonzero((err != ref).any(axis=1))
    assert bad.size == 0, "OSD output differs on shots %s" % bad[:10]
    # every output reproduces its syndrome (DEM-sampled syndromes are in the column space)
    Hd = np.asarray(H.todense(), dtype=np.int64)
    assert np.array_equal((err.astype(np.int64) @ Hd.T) % 2, synd)
    # posterior hand-off: the LLRs OSD saw are the oracle's BP posteriors, bit for bit
    b = int(np.flatnonzero(used_osd)[0])
    llr = dec.failed_llr(b).cpu().numpy()
    _, _, llr_ref, _ = g.bp(synd[b], prm)
    assert np.array_equal(llr.astype(np.float64), llr_ref)        # exact: the double-precision posteriors fit a float


def test_osd_inconsistent_and_rank_deficient(gpu):
    """Syndromes outside the column space: defined by the oracle (lowest-index pivot row), flagged in status."""
    H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")   # rank 1002 of 1008
    rng = np.random.default_rng(3)
    synd = (rng.random((64, H.shape[0])) < 0.15).astype(np.uint8)
    err, status, _ = _gpu_decode(H, pri, synd, 8, osd="osd_0")
    g, prm = _oracle(H, pri, 8, "osd_0")
    ref, flags = g.decode_batch(synd, prm)
    assert np.array_equal(err, ref)
    assert np.array_equal((status >> 18) & 1, flags[:, 3])
    assert flags[:, 3].sum() > 0


@pytest.mark.parametrize("name,code,cases", [
    ("bb72_custom_r6_p0.003", "bb72", ((3, 1, 20), (5, 3, 12), (8, 1, 30), (9, 2, 30))),
    ("hgp225_cardinal_r3_p0.01", "hgp225", ((3, 1, 15), (2, 1, 15))),
    ("bb72_custom_r2_xbasis_mixed", "bb72", ((2, 1, 15), (4, 1, 15), (3, 2, 15))),      # X-basis memory: hx / lx, four channel rates
])
def test_sliding_window_matches_reference_loop(gpu, name, code, cases):
    """The batched device driver against the REFERENCE's own per-shot loop (golden G5: reference
    sliding_window_circuit_mem driven with the oracle decoder as plug-in)."""
    import warnings
    from quits_amd.decoder import sliding_window_bposd_circuit_mem
    from quits_amd.dem import Circuit
    z = np.load(helpers.GOLD + "/loop/%s.npz" % name)
    shp = tuple(z["shape"])
    synd = np.unpackbits(z["syndromes"], axis=1)[:, :shp[1]]
    cd = helpers.code(code)
    if "xbasis" in name:
        cd = {"hz": cd["hx"], "lz": cd["lx"]}
    circ = Circuit(helpers.circuit_text(name))
    for (W, F, mi) in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pred = sliding_window_bposd_circuit_mem(synd, circ, cd["hz"], cd["lz"], W, F, max_iter=mi, osd_order=0,
                                                    bp_method="minimum_sum", schedule="parallel", osd_method="osd_0")
        assert pred.dtype == np.int64 and pred.shape == (shp[0], cd["lz"].shape[0])
        assert np.array_equal(pred, z["circ_W%dF%d_it%d_grid" % (W, F, mi)]), (W, F, mi)


@pytest.mark.parametrize("name,code,cases", [
    ("bb72_custom_r6_p0.003", "bb72", ((3, 1, 20), (5, 3, 12))),
    ("hgp225_cardinal_r3_p0.01", "hgp225", ((3, 1, 15), (2, 1, 15))),
])
def test_phenom_sliding_window_matches_reference_loop(gpu, name, code, cases):
    from quits_amd.decoder import sliding_window_bposd_phenom_mem
    z = np.load(helpers.GOLD + "/loop/%s.npz" % name)
    shp = tuple(z["shape"])
    synd = np.unpackbits(z["syndromes"], axis=1)[:, :shp[1]]
    cd = helpers.code(code)
    for (W, F, mi) in cases:
        pred = sliding_window_bposd_phenom_mem(synd, cd["hz"], cd["lz"], W, F, eff_error_rate_per_fault=0.03,
                                               max_iter=mi, osd_order=0, bp_method="minimum_sum",
                                               schedule="parallel", osd_method="osd_0")
        assert np.array_equal(pred, z["phen_W%dF%d_it%d_grid" % (W, F, mi)]), (W, F, mi)


def test_plugin_class_in_host_loop(gpu):
    """B1: the HIP decoder as a per-shot plug-in object (ldpc's constructor/decode surface)."""
    from quits_amd.decoder import BpOsdDecoder
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    synd, _, _ = orc.sample_dem(H, L, pri, seed=21, shot0=0, B=24)
    dec = BpOsdDecoder(H, channel_probs=pri, max_iter=15, bp_method="minimum_sum", schedule="parallel",
                       osd_method="osd_0", osd_order=0)
    g, prm = _oracle(H, pri, 15, "osd_0")
    ref, flags = g.decode_batch(synd, prm)
    for i in range(synd.shape[0]):
        e = dec.decode(synd[i].astype(int))
        assert e.dtype == np.int64 or e.dtype == int
        assert np.array_equal(e, ref[i])
        assert dec.converge == bool(flags[i, 0])


def test_unsupported_options_fail_loudly(gpu):
    from quits_amd.decoder import BpOsdDecoder
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    for kw in (dict(serial_schedule_order=list(range(H.shape[1]))), dict(random_schedule_seed=3),
               dict(osd_method="osd_cs", osd_order=65), dict(osd_method="osd_e", osd_order=16)):
        with pytest.raises(NotImplementedError):
            BpOsdDecoder(H, channel_probs=pri, max_iter=5, **{**dict(bp_method="minimum_sum", schedule="parallel",
                                                               osd_method="osd_0", osd_order=0), **kw})


def test_headline_scale_properties(gpu):
    """BASELINE config 3 size ([[144,12,12]], 12 rounds, single window) at a batch the oracle cannot follow:
    size-independent properties -- every output reproduces its syndrome, the zero syndrome maps to zero, decoding
    is invariant under permuting the shots, and the LER agrees with the oracle's within Monte-Carlo error."""
    import torch
    from quits_amd.decoder.device import (BatchDecoder, DemSampler, GF2Matrix, WindowGraph, count_mismatch)
    H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
    N = 20000
    det, obs = DemSampler(H, L, pri).sample(N, seed=99)
    g = WindowGraph(H, pri)
    dec = BatchDecoder(g, max_iter=50, osd_method="osd_0")
    bits, status = dec.decode(det)
    Hm, Lm = GF2Matrix(H), GF2Matrix(L)
    resyn = torch.empty((N, H.shape[0]), dtype=torch.uint8, device="cuda")
    Hm.xor_apply(bits, resyn, accumulate=False)
    assert torch.equal(resyn, det)
    pred = torch.zeros((N, L.shape[0]), dtype=torch.uint8, device="cuda")
    Lm.xor_apply(bits, pred, accumulate=True)
    fails = int(count_mismatch(pred, obs).item())
    perm = torch.randperm(N, device="cuda")
    bits2, status2 = dec.decode(det[perm].contiguous())
    assert torch.equal(bits2, bits[perm]) and torch.equal(status2 & 0xFFFFF, status[perm] & 0xFFFFF)
    # oracle LER on the first 600 of the same shots
    n_ref = 600
    go, prm = _oracle(H, pri, 50, "osd_0")
    ref, _ = go.decode_batch(det[:n_ref].cpu().numpy(), prm)
    assert np.array_equal(ref, np.unpackbits(bits[:n_ref].cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :H.shape[1]])
    p = fails / N
    assert 0.01 < p < 0.12, p


def _osd_only(H, pri, synd, llr):
    import torch
    from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits
    g = WindowGraph(H, pri)
    d = BatchDecoder(g, max_iter=1, osd_method="osd_0")
    bits, status = d.osd0(torch.from_numpy(np.ascontiguousarray(synd)).cuda(),
                          torch.from_numpy(np.ascontiguousarray(llr, dtype=np.float32)).cuda())
    return unpack_bits(bits, g.n).cpu().numpy(), status.cpu().numpy()


@pytest.mark.parametrize("case", ["ties", "few_values", "deep", "negzero"])
def test_osd_alone_on_crafted_llrs(gpu, case):
    """OSD-0 through qd_osd0_batch on soft information chosen to hit the rare paths of the kernel:
    ties   -- every LLR equal: > 1024 columns share one key, the order is purely by fault index (tier-by-index path);
    few_values -- LLRs from a 3-value alphabet: huge tie groups split across tiers;
    deep   -- random syndromes + random LLRs: elimination runs to full rank (1002 pivots -> Q planes spill to HBM);
    negzero -- +0.0 / -0.0 / tiny values must tie exactly like the oracle's '<' on doubles."""
    H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
    m, n = H.shape
    rng = np.random.default_rng({"ties": 1, "few_values": 2, "deep": 3, "negzero": 4}[case])
    B = 24
    synd, _, _ = orc.sample_dem(H, L, pri, seed=77, shot0=0, B=B)
    if case == "ties":
        llr = np.full((B, n), 2.5, np.float32)
    elif case == "few_values":
        llr = rng.choice(np.array([-1.0, 0.25, 3.0], np.float32), size=(B, n))
    elif case == "deep":
        synd = (rng.random((B, m)) < 0.3).astype(np.uint8)
        llr = rng.normal(size=(B, n)).astype(np.float32)
    else:
        llr = rng.choice(np.array([0.0, -0.0, 1e-30, -1e-30, 1.0], np.float32), size=(B, n))
    err, status = _osd_only(H, pri, synd, llr)
    g = orc.Graph(H, pri)
    for b in range(B):
        ref, st = g.osd0(synd[b], llr[b].astype(np.float64), stop_early=True)
        assert np.array_equal(err[b], ref), (case, b)
        assert ((status[b] >> 20) & 0xFFF) == min(st["pivots"], 4095)
        assert bool(status[b] & (1 << 18)) == st["inconsistent"]
    if case == "deep":
        assert ((status >> 20) & 0xFFF).max() > 900


@pytest.mark.parametrize("name,method,order,shots", [
    ("bb72_custom_r6_p0.003", "osd_cs", 1, 40),
    ("bb72_custom_r6_p0.003", "osd_cs", 7, 40),
    ("bb72_custom_r6_p0.003", "osd_e", 5, 40),
    ("hgp225_cardinal_r3_p0.01", "osd_cs", 3, 6),
    ("bb144_custom_r12_p0.003", "osd_cs", 2, 6),
    ("bb72_custom_r6_p0.003", "osd_cs", 64, 8),        # the largest orders the device takes: 2016 pairs / 4095 patterns over the first non-pivot columns
    ("bb72_custom_r6_p0.003", "osd_e", 12, 8),
    ("bb144_custom_r12_p0.003", "osd_e", 9, 4),
])
def test_higher_order_osd_bit_exact(gpu, name, method, order, shots):
    """OSD-CS / OSD-E on the device against the oracle with the same integer candidate costs: full-rank elimination,
    candidate sweep, ldpc's tie rule (earliest candidate wins)."""
    import torch
    from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits
    H, L, pri = helpers.dem_matrices(name)
    synd, _, _ = orc.sample_dem(H, L, pri, seed=31, shot0=0, B=shots)
    g, prm = _oracle(H, pri, 6, "osd_0")
    llr = np.zeros((shots, H.shape[1]), np.float32)
    for b in range(shots):
        _, _, l, _ = g.bp(synd[b], prm)
        llr[b] = l.astype(np.float32)
    wg = WindowGraph(H, pri)
    dec = BatchDecoder(wg, max_iter=6, osd_method=method, osd_order=order)
    bits, status = dec.osd0(torch.from_numpy(synd).cuda(), torch.from_numpy(llr).cuda())
    err = unpack_bits(bits, wg.n).cpu().numpy()
    improved = 0
    for b in range(shots):
        ref, st = g.osd_w(synd[b], llr[b].astype(np.float64), method, order, fixed=True)
        assert np.array_equal(err[b], ref), (b, st)
        improved += st["winner"] > 0
    assert improved > 0, "no candidate ever beat OSD-0: the sweep is not exercised"
    # end to end (BP + OSD-CS) through the batch decoder
    bits2, status2 = dec.decode(torch.from_numpy(synd).cuda())
    ref2, flags2 = g.decode_batch(synd, orc.make_params("minimum_sum", "parallel", 6, method, order, 1.0, orc.FORM_LDPC_F64))
    assert np.array_equal(unpack_bits(bits2, wg.n).cpu().numpy(), ref2)


# ---- the one-message-per-edge kernel (csrc/bp_general.hip): every bp_method x schedule pair the reference wrapper can
# request, against the oracle's ldpc-form in float (bp_parallel_edge_f32 / bp_serial_edge_f32), bit for bit.
def _gpu_decode_general(H, pri, synd, method, schedule, max_iter, osd="osd_off", order=0, alpha=1.0):
    import torch
    from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits
    g = WindowGraph(H, pri)
    d = BatchDecoder(g, bp_method=method, schedule=schedule, max_iter=max_iter, osd_method=osd, osd_order=order,
                     ms_scaling_factor=alpha, edge_messages=True)
    bits, status = d.decode(torch.from_numpy(np.ascontiguousarray(synd)).cuda())
    return unpack_bits(bits, g.n).cpu().numpy(), status.cpu().numpy()


@pytest.mark.parametrize("name,shots,method,schedule,max_iter,alpha", [
    ("bb72_custom_r6_p0.003", 300, "product_sum", "parallel", 20, 1.0),
    ("bb72_custom_r6_p0.003", 300, "product_sum", "serial", 6, 1.0),
    ("bb72_custom_r6_p0.003", 300, "minimum_sum", "serial", 6, 1.0),
    ("bb72_custom_r6_p0.003", 300, "minimum_sum", "serial", 6, 0.0),
    ("bb72_custom_r6_p0.003", 300, "minimum_sum", "parallel", 20, 0.75),
    ("bb72_custom_r6_p0.003", 300, "minimum_sum", "parallel", 20, 1.0),     # on the LLR grid: equals ldpc's double arithmetic
    ("hgp225_cardinal_r3_p0.01", 130, "product_sum", "serial", 3, 1.0),
    ("hgp225_cardinal_r3_p0.01", 130, "product_sum", "parallel", 10, 1.0),
    ("bb144_custom_r12_p0.003", 70, "product_sum", "serial", 2, 1.0),      # the reference wrapper's defaults (bposd.py:54)
    ("bb72_custom_r6_p0.003", 300, "product_sum", "serial", 12, 1.0),      # serial schedule in four launches (bounds 3, 6, 10), survivors packed in between
    ("bb72_custom_r6_p0.003", 300, "minimum_sum", "serial", 12, 0.0),      # ... with the iteration-dependent scaling carried across the launches
])
def test_general_bp_bit_exact(gpu, name, shots, method, schedule, max_iter, alpha):
    H, L, pri = helpers.dem_matrices(name)
    synd, _, _ = orc.sample_dem(H, L, pri, seed=21, shot0=0, B=shots)
    synd[3] = 0
    err, status = _gpu_decode_general(H, pri, synd, method, schedule, max_iter, alpha=alpha)
    go, form = orc.device_arithmetic(H, pri, method, schedule, max_iter, alpha)
    if form == orc.FORM_COMPRESSED_F32:
        form = orc.FORM_LDPC_F32                   # edge_messages: ldpc's update order in float
    ref, flags = go.decode_batch(synd, orc.make_params(method, schedule, max_iter, "osd_off", 0, alpha, form))
    assert np.array_equal((status >> 16) & 1, flags[:, 0]), "convergence flags differ"
    assert np.array_equal(status & 0x3FFF, flags[:, 1]), "iteration counts differ"
    assert np.array_equal(err, ref), "hard decisions differ"
    assert status[3] & (1 << 19) and not err[3].any()


@pytest.mark.parametrize("method,schedule,osd,order,max_iter", [
    ("product_sum", "serial", "osd_cs", 0, 2),       # bposd.py:54 defaults: osd_cs with osd_order 0 is OSD-0
    ("product_sum", "serial", "osd_cs", 3, 2),
    ("product_sum", "parallel", "osd_0", 0, 8),
    ("minimum_sum", "serial", "osd_e", 4, 3),
])
def test_general_bposd_bit_exact(gpu, method, schedule, osd, order, max_iter):
    """BP in the general kernel, its posteriors handed to the OSD kernels (through the [fault][shot] -> fail-slot rows
    transpose): predictions equal the oracle's ldpc-form float decoder followed by its OSD."""
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    synd, _, _ = orc.sample_dem(H, L, pri, seed=33, shot0=0, B=400)
    err, status = _gpu_decode_general(H, pri, synd, method, schedule, max_iter, osd=osd, order=order)
    prm = orc.make_params(method, schedule, max_iter, osd, order, 1.0, orc.FORM_LDPC_F32)
    ref, flags = orc.Graph(H, pri).decode_batch(synd, prm)
    used_osd = (status >> 17) & 1
    assert np.array_equal(used_osd, 1 - flags[:, 0])
    assert used_osd.sum() > 10, "test does not exercise OSD"
    assert np.array_equal(err, ref)


def test_general_serial_staged_launches_equal_one_launch(gpu, monkeypatch):
    """The serial schedule in several launches (GenStage: the shots still running are packed into full wavefronts after iterations 3, 6, 10, ...)
    returns what ONE launch returns -- decisions, iteration counts, flags, and the OSD results of the shots BP could not finish (their posteriors
    come from the last launch's workspace, by its columns) -- for other bounds too, for a batch decoded in several workspace chunks, and equals the oracle."""
    import torch
    from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    det, _ = DemSampler(H, L, pri).sample(1500, seed=18)
    det[5] = 0
    g = WindowGraph(H, pri)
    for method, osd, order in (("product_sum", "osd_cs", 2), ("minimum_sum", "osd_0", 0)):
        kw = dict(bp_method=method, schedule="serial", max_iter=9, osd_method=osd, osd_order=order, edge_messages=True)
        monkeypatch.setenv("QD_GEN_STAGES", "0")
        one = BatchDecoder(g, **kw)
        bits_1, st_1 = one.decode(det)
        assert ((st_1 >> 17) & 1).sum() > 20, "test does not exercise the hand-over to OSD"
        for stages in (None, "1,2,3,4,5,6,7", "4"):
            if stages is None:
                monkeypatch.delenv("QD_GEN_STAGES")
            else:
                monkeypatch.setenv("QD_GEN_STAGES", stages)
            d = BatchDecoder(g, **kw)
            bits, st = d.decode(det)
            assert torch.equal(bits, bits_1) and torch.equal(st, st_1), (method, stages)
        # shots that do not converge (random syndromes): packing returns nothing, the decoder notices from the counts of its first call and runs
        # the next ones in one launch -- same results before and after the switch
        bad = (torch.rand((4352, det.shape[1]), device=det.device) < 0.25).to(torch.uint8)
        bits_b, st_b = one.decode(bad)
        for rep in range(3):
            bits, st = d.decode(bad)
            torch.cuda.synchronize()
            assert torch.equal(bits, bits_b) and torch.equal(st, st_b), (method, "no convergence", rep)
        bits, st = d.decode(det)
        assert torch.equal(bits, bits_1) and torch.equal(st, st_1), (method, "after the switch")
        monkeypatch.setenv("QD_GENERAL_WS_GB", "0.02")          # ~ 256-shot chunks (two message planes of them)
        monkeypatch.delenv("QD_GEN_STAGES", raising=False)
        d = BatchDecoder(g, **kw)
        bits, st = d.decode(det)
        assert torch.equal(bits, bits_1) and torch.equal(st, st_1), (method, "chunks")
        monkeypatch.delenv("QD_GENERAL_WS_GB")
        from quits_amd.decoder.device import unpack_bits
        synd = det[:200].cpu().numpy()
        go, form = orc.device_arithmetic(H, pri, method, "serial", 9, 1.0)
        form = orc.FORM_LDPC_F32 if form == orc.FORM_COMPRESSED_F32 else form
        ref, flags = go.decode_batch(synd, orc.make_params(method, "serial", 9, osd, order, 1.0, form))
        assert np.array_equal(unpack_bits(bits_1[:200], g.n).cpu().numpy(), ref)
        assert np.array_equal((st_1[:200].cpu().numpy() >> 16) & 1, flags[:, 0])


def test_general_chunking_and_compressed_agreement(gpu, monkeypatch):
    """A batch larger than the workspace chunk is decoded chunk by chunk with identical results; and flooding min-sum in
    the edge form (ldpc's prefix sums, messages in HBM) returns the SAME BITS as the compressed LDS kernel ("total minus
    own"): on the LLR grid both compute exactly.  With QD_FLAG_RAW_LLR (round-1 float arithmetic) they only nearly agree."""
    import torch
    from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    det, _ = DemSampler(H, L, pri).sample(1500, seed=8)
    g = WindowGraph(H, pri)
    a = BatchDecoder(g, max_iter=20, osd_method="osd_0", edge_messages=True)
    bits_a, st_a = a.decode(det)
    monkeypatch.setenv("QD_GENERAL_WS_GB", "0.02")          # ~ 256-shot chunks
    b = BatchDecoder(g, max_iter=20, osd_method="osd_0", edge_messages=True)
    bits_b, st_b = b.decode(det)
    assert torch.equal(bits_a, bits_b) and torch.equal(st_a, st_b)
    c = BatchDecoder(g, max_iter=20, osd_method="osd_0")
    bits_c, st_c = c.decode(det)
    assert not (st_c & (3 << 14)).any()                      # no shot left the fine grid
    assert torch.equal(bits_a, bits_c) and torch.equal(st_a, st_c)
    r1 = BatchDecoder(g, max_iter=20, osd_method="osd_0", raw_llr=True)
    r2 = BatchDecoder(g, max_iter=20, osd_method="osd_0", raw_llr=True, edge_messages=True)
    assert r1.info()["llr_grid_bits"] == -1
    same = (r1.decode(det)[0] == r2.decode(det)[0]).all(dim=1).float().mean().item()
    assert 0.9 < same, same


def test_reference_defaults_run_on_device(gpu):
    """`sliding_window_bposd_circuit_mem` with the reference's own default options (product_sum, serial, osd_cs,
    max_iter=2, osd_order=0; bposd.py:54) against the oracle's sliding-window loop in the same float arithmetic."""
    from quits_amd.decoder import sliding_window_bposd_circuit_mem
    from quits_amd.dem import Circuit
    name = "bb72_custom_r6_p0.003"
    cd = helpers.code("bb72")
    H, L, pri = helpers.dem_matrices(name)
    det, obs, _ = orc.sample_dem(H, L, pri, seed=4, shot0=0, B=200)
    pred = sliding_window_bposd_circuit_mem(det, Circuit(helpers.circuit_text(name)), cd["hz"], cd["lz"], 3, 1)
    wins = helpers.window_set(name, 3, 1)
    nz = cd["hz"].shape[0]
    for k, w in enumerate(wins):
        w["row0"] = k * nz
    prm = orc.make_params("product_sum", "serial", 2, "osd_cs", 0, 1.0, orc.FORM_LDPC_F32)
    ref, stats = orc.sliding_window_decode(wins, nz, det, prm)
    assert pred.dtype == np.int64 and np.array_equal(pred, ref.astype(np.int64))
    assert stats["osd_calls"] > 0


def test_codecap_driver_on_device(gpu):
    """`get_codecap_pL` with the HIP plug-in (all trials in one batched decode) against the reference's function run with
    the oracle's float forms (golden G8): identical logical error rates."""
    import json, os, types
    from quits_amd.decoder import BpOsdDecoder
    from quits_amd.simulation import get_codecap_pL
    seen = 0
    for ent in json.load(open(os.path.join(helpers.GOLD, "codecap.json"))):
        if ent["form"] == "f64":      # "grid" (flooding min-sum) and "f32" (product-sum) are what the device computes
            continue
        cd = helpers.code(ent["code"])
        cobj = types.SimpleNamespace(hz=cd["hz"], hx=cd["hx"], lz=cd["lz"], lx=cd["lx"])
        pl = get_codecap_pL(cobj, ent["p"], ent["trials"], BpOsdDecoder, dict(ent["opts"], error_rate=ent["p"]),
                            basis=ent["basis"], seed=ent["seed"])
        assert pl == ent["pL"], ent
        seen += 1
    assert seen == 4


def test_randomised_parity_sweep(gpu):
    """Random sparse check matrices (4..2600 detectors, column weight 1..6), equal / log-uniform / discrete priors, sampled and
    arbitrary syndromes, every bp_method x schedule x osd_method/order the device path offers: decisions, convergence flags,
    iteration counts, OSD use, pivot counts and the inconsistent flag equal the oracle's (tools/stress_parity.py; 2000 further
    cases were run once for profiles/r01_stress_parity.txt)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_parity
    ok, skipped = stress_parity.run(80, seed=20260929)
    assert ok >= 70


def test_every_window_shape(gpu):
    """All (W, F) with 1 <= F <= W <= R + 3 for the [[72,12,6]] circuit (including F = W and the whole-history branch
    W > R + 2): the batched device driver and the oracle's restatement of the reference loop give identical predictions."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_windows
    assert stress_windows.run(24, opts=(("minimum_sum", "parallel", 10, "osd_0", 0),)) == 45


# ---- BASELINE.json configs[3] and configs[4] ("configs_untested") ------------------------------------------
def _circuit_dem(name, p_from=None, p_to=None):
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.dem import Circuit
    text = helpers.circuit_text(name) if p_to is None else helpers.circuit_text_at_p(name, p_from, p_to)
    circ = Circuit(text)
    return circ, detector_error_model_to_matrix(circ)


@pytest.mark.parametrize("W,F,opts", [
    (8, 1, dict(bp_method="minimum_sum", schedule="parallel", max_iter=20, osd_method="osd_0", osd_order=0)),      # single window
    (8, 1, dict(bp_method="minimum_sum", schedule="parallel", max_iter=20, osd_method="lsd_cs", osd_order=1)),
    (3, 1, dict(bp_method="minimum_sum", schedule="parallel", max_iter=20, osd_method="osd_0", osd_order=0)),      # six windows, hand-off
    (3, 1, dict(bp_method="product_sum", schedule="serial", max_iter=4, osd_method="osd_cs", osd_order=1)),        # the per-edge BP kernel
])
def test_pipelined_chunks_equal_single_stream(gpu, W, F, opts):
    """A call of two or more chunks runs the post-processing (OSD / LSD, acc ^= L e, the hand-off U e) on a second stream
    beside the BP of the other chunk of a pair, with a second set of decoders (sliding_window.py _decode_pipelined_impl).
    Predictions and status words must equal the single-stream path's, shot for shot -- three chunks, the last one ragged, twice
    in a row (workspaces, buffers and streams reused), with work queued on the caller's stream before (the syndromes are
    sampled there) and after (the comparison)."""
    import torch
    from quits_amd.decoder.device import DemSampler
    from quits_amd.decoder.sliding_window import build_circuit_plan
    name, R = "bb72_custom_r6_p0.003", 6
    circ, (H, L, pri) = _circuit_dem(name)
    hz = helpers.code("bb72")["hz"]
    plan = build_circuit_plan(circ, hz, W, F, R, dict(opts), dict(opts))
    assert len(plan.windows) == (1 if W == 8 else 6)
    assert plan.pipeline == (opts["bp_method"] == "minimum_sum")      # on by default for the LDS kernels only
    plan.pipeline = True
    N = 2 * plan.chunk + 12345
    sampler = DemSampler(H, L, pri)
    for rep in range(2):
        det, obs = sampler.sample(N, seed=77 + rep)
        st_p, st_s = [], []
        pred_p = plan.decode(det, st_p)
        plan.pipeline = False
        pred_s = plan.decode(det, st_s)
        plan.pipeline = True
        assert len(st_p) == len(st_s) == 3 * len(plan.windows)
        assert torch.equal(pred_p, pred_s)
        by_p = {k: torch.cat([t for kk, t in st_p if kk == k]) for k in range(len(plan.windows))}
        by_s = {k: torch.cat([t for kk, t in st_s if kk == k]) for k in range(len(plan.windows))}
        for k in by_p:
            assert torch.equal(by_p[k], by_s[k]), k
        frac_post = float(((by_p[0] >> 17) & 1).float().mean())
        assert 0.005 < frac_post < 0.98, frac_post          # the post-processor is exercised
    # several groups of lanes (plans of more than one window run three lanes: groups of 3 + 3 + 3 + 2 chunks here, 2 + 2 + ... on two lanes)
    assert plan.lanes == (2 if W == 8 else 3)
    plan.chunk //= 5
    pred_p = plan.decode(det)
    plan.pipeline = False
    assert torch.equal(pred_p, plan.decode(det)) and torch.equal(pred_p, pred_s)


def test_public_call_plan_cache_and_streamed_host_samples(gpu, monkeypatch):
    """the drop-in call sliding_window_bposd_circuit_mem(host ndarray, circuit, ...) keeps its plan (DEM, windows,
    graphs, workspaces) between calls -- the reference is called once per experiment point -- and streams the host samples through
    pinned pieces.  A cached plan returns what a fresh one returns; another circuit or another option misses the cache; pieces
    smaller than the batch (three of them, the last ragged) and bool / uint8 / int64 / torch inputs all give the same int64 array,
    equal to the device-resident plan's."""
    import torch
    from quits_amd.decoder import sliding_window_bposd_circuit_mem
    from quits_amd.decoder import sliding_window as sw
    from quits_amd.decoder.device import DemSampler
    from quits_amd.dem import Circuit
    name, R = "bb72_custom_r6_p0.003", 6
    circ, (H, L, pri) = _circuit_dem(name)
    cd = helpers.code("bb72")
    hz, lz = cd["hz"], cd["lz"]
    kw = dict(max_iter=20, osd_order=0, bp_method="minimum_sum", schedule="parallel", osd_method="osd_0")
    det, obs = DemSampler(H, L, pri).sample(3000, seed=5)
    det_h = det.cpu().numpy()
    sw.plan_cache_clear()
    monkeypatch.setenv("QD_PLAN_CACHE", "4")
    ref = sw.build_circuit_plan(circ, hz, 3, 1, R, dict(kw), dict(kw)).decode(det).cpu().numpy().astype(np.int64)
    a = sliding_window_bposd_circuit_mem(det_h.astype(np.bool_), circ, hz, lz, 3, 1, **kw)
    assert a.dtype == np.int64 and np.array_equal(a, ref)
    assert sw.plan_cache_info()["misses"] == 1 and sw.plan_cache_info()["hits"] == 0
    # same arguments (the circuit as fresh text, hz as a copy): the cached plan, same answer; every input type
    for samples in (det_h, det_h.astype(np.int64), det_h.astype(np.bool_), torch.from_numpy(det_h), det):
        b = sliding_window_bposd_circuit_mem(samples, Circuit(str(circ)), hz.copy(), lz, 3, 1, **kw)
        assert b.dtype == np.int64 and np.array_equal(b, ref)
    info = sw.plan_cache_info()
    assert info["misses"] == 1 and info["hits"] == 5 and info["size"] == 1
    # pieces of 1024 shots through the staging buffers (3000 = 1024 + 1024 + 952)
    plan = next(iter(sw._CACHE.values()))
    plan.chunk, plan.host_piece, plan._stage = 1024, 1024, None
    assert np.array_equal(plan.decode_host(det_h.astype(np.bool_)), ref)
    assert np.array_equal(plan.decode_host(det_h[:10]), ref[:10]) and plan.decode_host(det_h[:0]).shape == (0, ref.shape[1])
    # pieces of two or more chunks: ONE chain of the two-lane pipeline across the pieces (3000 = 4 x 256 | 4 x 256 | 3 x 256 + 184;
    # 2900 and 2400 leave a last piece of less than two chunks / less than one, chained like the others), and the same with the chain switched off
    assert plan.pipeline
    plan.chunk, plan.host_piece, plan._stage = 256, 1024, None
    assert np.array_equal(plan.decode_host(det_h), ref)
    plan.chunk, plan.host_piece, plan._stage = 500, 1000, None
    assert np.array_equal(plan.decode_host(det_h[:2900]), ref[:2900])
    assert np.array_equal(plan.decode_host(det_h[:2400]), ref[:2400])
    monkeypatch.setenv("QD_NO_HOST_CHAIN", "1")
    assert np.array_equal(plan.decode_host(det_h[:2900]), ref[:2900])
    monkeypatch.delenv("QD_NO_HOST_CHAIN")
    assert np.array_equal(plan.decode(det).cpu().numpy().astype(np.int64), ref)          # (the plan's device-resident call after a chain)
    # another option / another window shape / another circuit: misses
    sliding_window_bposd_circuit_mem(det_h, circ, hz, lz, 3, 1, **dict(kw, max_iter=21))
    sliding_window_bposd_circuit_mem(det_h, circ, hz, lz, 5, 3, **kw)
    other = helpers.circuit_text_at_p(name, 0.003, 0.002)
    c2 = sliding_window_bposd_circuit_mem(det_h, Circuit(other), hz, lz, 3, 1, **kw)
    info = sw.plan_cache_info()
    assert info["misses"] == 4 and info["size"] == 4
    assert not np.array_equal(c2, ref)                     # other priors: a different decoder, not the cached one
    sw.plan_cache_clear()


@pytest.mark.parametrize("p,shots", [(0.001, 400), (0.002, 300), (0.004, 200), (0.005, 200), (0.006, 200)])
def test_config3_p_sweep_points_bit_exact(gpu, p, shots):
    """configs[3]: the [[144,12,12]] single window at every point of the p-sweep but the headline's (p = 1e-3: BP converges on
    ~98 % of the shots; p = 6e-3: practically every shot goes through OSD), device against the oracle's double-precision
    ldpc-order decoder."""
    circ, (H, L, pri) = _circuit_dem("bb144_custom_r12_p%g" % p)
    synd, obs, _ = orc.sample_dem(H, L, pri, seed=606, shot0=0, B=shots)
    err, status, dec = _gpu_decode(H, pri, synd, 50, osd="osd_0")
    g, prm = _oracle(H, pri, 50, "osd_0")
    ref, flags, grid = g.decode_batch(synd, prm, return_grid=True)
    assert np.array_equal((status >> 16) & 1, flags[:, 0]) and np.array_equal(status & 0x3FFF, flags[:, 1])
    assert np.array_equal((status >> 14) & 1, (grid[:, 0] != g.grid[0]).astype(int))
    assert np.array_equal((status >> 20) & 0xFFF, np.minimum(flags[:, 2], 4095)), "pivot counts differ"
    assert np.array_equal(err, ref)
    conv = flags[:, 0].mean()
    assert (conv > 0.9) if p == 0.001 else (conv < 0.05 if p == 0.006 else 0.0 <= conv <= 1.0), conv


@pytest.mark.parametrize("osd,order,shots,max_iter", [("osd_0", 0, 128, 30), ("osd_cs", 1, 64, 30)])
def test_config4_qlp_sliding_window_bit_exact(gpu, osd, order, shots, max_iter):
    """configs[4]: QLP [[1020,136]], cardinal circuit, R = 20, sliding window W = 3 F = 1 (20 windows of 1350 x 18900: 77-wide
    checks -> separate sign words, 22-word OSD rows, the 128-register BP instantiation), OSD-0 and OSD-CS order 1, at
    p = 1e-3 (at the fixture's p = 3e-3 the code is above threshold).  Device driver against the oracle's loop."""
    from quits_amd.decoder import sliding_window_bposd_circuit_mem
    from quits_amd.decoder.base import spacetime, window_count
    name, R = "qlp1020_cardinal_r20_p0.003", 20
    circ, (H, L, pri) = _circuit_dem(name, 0.003, 0.001)
    cd = helpers.code("qlp1020")
    hz, lz = cd["hz"], cd["lz"]
    nz = hz.shape[0]
    det, obs, _ = orc.sample_dem(H, L, pri, seed=44, shot0=0, B=shots)
    pred = sliding_window_bposd_circuit_mem(det, circ, hz, lz, 3, 1, max_iter=max_iter, osd_order=order,
                                            bp_method="minimum_sum", schedule="parallel", osd_method=osd)
    ncr, _, _ = window_count(R, 3, 1)
    checks, commits, priors, updates = spacetime(circ, hz, 3, 1, ncr)
    assert len(checks) == 20 and checks[1].shape == (1350, 18900)
    wins = [{"H": checks[k], "L": commits[k], "priors": priors[k], "U": updates[k] if k < ncr else None, "row0": k * nz}
            for k in range(len(checks))]
    ref, stats = helpers.oracle_sliding_window_parallel(wins, nz, det, ("minimum_sum", "parallel", max_iter, osd, order, 1.0, orc.FORM_LDPC_F64),
                                                        device_grid=True)
    assert stats["osd_calls"] > shots, "OSD is not exercised"
    assert pred.shape == (shots, lz.shape[0]) and np.array_equal(pred, ref.astype(np.int64))


@pytest.mark.parametrize("method,order,shots", [("lsd_0", 0, 64), ("lsd_cs", 1, 32)])
def test_config4_qlp_sliding_window_bplsd_bit_exact(gpu, method, order, shots):
    """configs[4]'s code and window plan through BP-LSD (sliding_window_bplsd_circuit_mem; 20 windows of 1350 x 18900: the LSD
    kernel's 24-checks-per-lane instantiation, 77-wide rows = two ELL chunks per rescan), LSD-0 and the order the reference's
    own BP-LSD calls use, against the oracle's loop."""
    from quits_amd.decoder import sliding_window_bplsd_circuit_mem
    from quits_amd.decoder.base import spacetime, window_count
    name, R = "qlp1020_cardinal_r20_p0.003", 20
    circ, (H, L, pri) = _circuit_dem(name, 0.003, 0.001)
    cd = helpers.code("qlp1020")
    hz, lz = cd["hz"], cd["lz"]
    nz = hz.shape[0]
    det, obs, _ = orc.sample_dem(H, L, pri, seed=45, shot0=0, B=shots)
    pred = sliding_window_bplsd_circuit_mem(det, circ, hz, lz, 3, 1, max_iter=30, lsd_order=order, bp_method="minimum_sum",
                                            schedule="parallel", lsd_method=method)
    ncr, _, _ = window_count(R, 3, 1)
    checks, commits, priors, updates = spacetime(circ, hz, 3, 1, ncr)
    wins = [{"H": checks[k], "L": commits[k], "priors": priors[k], "U": updates[k] if k < ncr else None, "row0": k * nz}
            for k in range(len(checks))]
    ref, stats = helpers.oracle_sliding_window_parallel(wins, nz, det, ("minimum_sum", "parallel", 30, method, order, 1.0, orc.FORM_LDPC_F64),
                                                        device_grid=True)
    assert stats["osd_calls"] > shots, "LSD is not exercised"
    assert np.array_equal(pred, ref.astype(np.int64))


@pytest.mark.parametrize("osd,order", [("lsd_0", 0), ("lsd_cs", 1), ("osd_cs", 1), ("osd_e", 6)])
def test_headline_window_2048_shots_other_postprocessors(gpu, osd, order):
    """The post-processors other than OSD-0 (whose 200 000-shot golden lives in tests/golden/ler) at the headline window on
    2048 shots: every correction, convergence flag and iteration count against the oracle (tools/scale_parity.py at 16 384)."""
    from quits_amd.decoder.device import DemSampler
    H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
    shots = 2048
    det, obs = DemSampler(H, L, pri).sample(shots, seed=9)
    synd = det.cpu().numpy()
    err, st, dec = _gpu_decode(H, pri, synd, 50, osd=osd, order=order)
    ref, flags = helpers.oracle_decode_batch_parallel(H, pri, synd, ("minimum_sum", "parallel", 50, osd, order, 1.0, orc.FORM_LDPC_F64),
                                                      device_grid_max_iter=50)
    assert np.array_equal((st >> 16) & 1, flags[:, 0]) and np.array_equal(st & 0x3FFF, flags[:, 1])
    assert ((st >> 17) & 1).sum() > 400
    bad = np.flatnonzero((err != ref).any(axis=1))
    assert bad.size == 0, bad[:10]


def test_all_detectors_circuit_decodes(gpu):
    """a circuit built with CircuitBuildOptions(get_all_detectors=True, noisy_zeroth_round=False,
    noisy_final_meas=True): X and Z detectors in one record.  The reference's window slicer assumes one detector type per
    round, so such a record is decoded as one BP-OSD problem over the whole DEM (ldpc's surface, B1): device against the
    oracle, every output reproduces its syndrome, and using the X detectors too does not hurt the logical error rate."""
    from quits_amd.decoder import BpOsdDecoder
    circ, (H, L, pri) = _circuit_dem("bb72_custom_r2_alldet_p0.003")
    assert H.shape[0] == 216
    synd, obs, _ = orc.sample_dem(H, L, pri, seed=9, shot0=0, B=300)
    err, status, _ = _gpu_decode(H, pri, synd, 30, osd="osd_0")
    g, prm = _oracle(H, pri, 30, "osd_0")
    ref, flags = g.decode_batch(synd, prm)
    assert np.array_equal(err, ref) and np.array_equal((status >> 16) & 1, flags[:, 0])
    Hd = np.asarray(H.todense(), dtype=np.int64)
    assert np.array_equal((err.astype(np.int64) @ Hd.T) % 2, synd)
    Ld = np.asarray(L.todense(), dtype=np.int64)
    fails = ((err.astype(np.int64) @ Ld.T) % 2 != obs).any(axis=1).mean()
    assert fails < 0.1, fails
    dec = BpOsdDecoder(H, channel_probs=pri, max_iter=30, bp_method="minimum_sum", schedule="parallel", osd_method="osd_0", osd_order=0)
    assert np.array_equal(dec.decode(synd[5].astype(int)), ref[5])


def test_ler_agreement_with_double_precision_oracle_2e5_shots(gpu):
    """north_star: logical error rate within Monte-Carlo error of the reference arithmetic on identical syndromes.  Golden: the
    oracle's DOUBLE-precision ldpc-order decoder on the first 200 000 Philox-sampled shots (seed 1) of the headline configuration,
    on the exact channel LLRs (`ldpc_f64`) and on the 2^-11 grid (`ldpc_f64_q11`); made by tools/ler_forms.py in 20 core-minutes.
      * the device's predictions equal the grid column on every one of the 200 000 shots (iteration counts too);
      * its logical error rate is within one sigma of the exact-LLR column, paired (McNemar) and unpaired.
    (10^6 and 2 x 10^6 shot versions of the same comparison: profiles/r02_ler_forms_*.json.)"""
    import torch
    from quits_amd.decoder.device import BatchDecoder, DemSampler, GF2Matrix, WindowGraph
    z = np.load(helpers.GOLD + "/ler/bb144_p0.003_seed1_first200k.npz")
    N = len(z["obs"])
    H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
    smp, g, Lm = DemSampler(H, L, pri), WindowGraph(H, pri), GF2Matrix(L)
    dec = BatchDecoder(g, max_iter=50, osd_method="osd_0")
    assert dec.info()["llr_grid_bits"] == 11
    w = (1 << torch.arange(L.shape[0], device="cuda")).to(torch.int32)
    pred, its, obs = [], [], []
    for c0 in range(0, N, 50000):
        det, ob = smp.sample(50000, seed=1, shot0=c0)
        bits, status = dec.decode(det)
        assert not (status & (3 << 14)).any()
        p = torch.zeros((50000, L.shape[0]), dtype=torch.uint8, device="cuda")
        Lm.xor_apply(bits, p, accumulate=False)
        pred.append((p.to(torch.int32) * w).sum(1).cpu().numpy().astype(np.uint16))
        obs.append((ob.to(torch.int32) * w).sum(1).cpu().numpy().astype(np.uint16))
        its.append((status & 0x3FFF).cpu().numpy().astype(np.uint8))
    pred, obs, its = np.concatenate(pred), np.concatenate(obs), np.concatenate(its)
    assert np.array_equal(obs, z["obs"]), "the device sampler and the oracle's sampler disagree"
    assert np.array_equal(pred, z["ldpc_f64_q11_pred"]) and np.array_equal(its, z["ldpc_f64_q11_iters"])
    f_dev, f_ref = pred != obs, z["ldpc_f64_pred"] != obs
    n_dev, n_ref = int(f_dev.sum()), int(f_ref.sum())
    p_ref = n_ref / N
    sigma = np.sqrt(p_ref * (1 - p_ref) / N)
    b, c = int((f_dev & ~f_ref).sum()), int((~f_dev & f_ref).sum())
    assert abs(n_dev - n_ref) / N <= sigma, (n_dev, n_ref, sigma * N)
    assert abs(b - c) <= np.sqrt(b + c), (b, c)


def test_ler_agreement_product_sum_serial_osdcs_1e5_shots(gpu):
    """The reference wrapper's own settings (bposd.py:54 defaults + the notebooks' max_iter = 10, osd_order = 1: product_sum,
    serial, osd_cs) at the headline window: the device's float path against ldpc's arithmetic (oracle: double, libm, bp.hpp's
    update order; tests/golden/ler/bb144_ps_serial_osdcs1_seed1_part*.npz, made by tools/ler_productsum.py) on the same
    100 000 Philox shots, paired.  Since the device keeps e^-|x| in place of tanh(|x| / 2) (qd_math.h: no clamp at 1 - 2^-24)
    the two agree almost shot for shot: the same prediction on more than 99.9 % of the shots, a handful of discordant failures,
    failure counts within a quarter sigma.  (The clamped float form of rounds 1-3 sat at -0.5 sigma, McNemar z = -2.7.)"""
    import glob
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(helpers.GOLD.rstrip("/")), "..", "tools"))
    import ler_productsum as lp
    paths = sorted(glob.glob(os.path.join(helpers.GOLD, "ler", "bb144_ps_serial_osdcs1_seed1_part*.npz")))[:2]
    assert len(paths) == 2
    fd, fr = [], []
    for path in paths:
        z = np.load(path)
        meta = json.loads(bytes(z["meta"]).decode())
        obs, pred, conv = lp.device_predictions(meta["seed"], meta["shot0"], meta["shots"])
        assert np.array_equal(obs, z["obs"]), "device sampler and oracle sampler disagree"
        fd.append(pred != obs)
        fr.append(z["pred"] != z["obs"])
        assert (pred == z["pred"]).mean() > 0.999           # float vs double: the same prediction on all but a few shots in ten thousand
    r = lp.paired(np.concatenate(fd), np.concatenate(fr))
    assert r["shots"] == 100000
    assert sum(r["discordant"]) <= 20 and abs(r["mcnemar_z"]) <= 3.0 and abs(r["delta_in_sigma"]) <= 0.25, r


# ---- BP-LSD (quits/decoder/bplsd.py; csrc/lsd_kernels.hip) -------------------------------------------------------------------
@pytest.mark.parametrize("name,shots,max_iter", [
    ("bb72_custom_r6_p0.003", 1500, 8),
    ("hgp225_cardinal_r3_p0.01", 150, 10),
    ("bb144_custom_r12_p0.003", 400, 30),
])
def test_bplsd_bit_exact(gpu, name, shots, max_iter):
    """BP on the device, then LSD-0 (cluster growth + on-the-fly elimination, one wavefront per shot) against the oracle's
    restatement: corrections, pivot counts and the inconsistent flag, bit for bit; every output reproduces its syndrome."""
    H, L, pri = helpers.dem_matrices(name)
    synd, _, _ = orc.sample_dem(H, L, pri, seed=12, shot0=0, B=shots)
    synd[1] = 0
    err, status, dec = _gpu_decode(H, pri, synd, max_iter, osd="lsd_0")
    g, prm = _oracle(H, pri, max_iter, "lsd_0")
    ref, flags = g.decode_batch(synd, prm)
    used = (status >> 17) & 1
    assert np.array_equal(used, 1 - flags[:, 0]) and used.sum() > 10
    assert np.array_equal((status >> 20) & 0xFFF, np.minimum(flags[:, 2], 4095)), "pivot counts differ"
    assert np.array_equal((status >> 18) & 1, flags[:, 3])
    bad = np.flatnonzero((err != ref).any(axis=1))
    assert bad.size == 0, "LSD output differs on shots %s" % bad[:10]
    Hd = np.asarray(H.todense(), dtype=np.int64)
    assert np.array_equal((err.astype(np.int64) @ Hd.T) % 2, synd)


@pytest.mark.parametrize("name,shots,max_iter,method,order", [
    ("bb72_custom_r6_p0.003", 1500, 8, "lsd_cs", 1),          # the order every BP-LSD call of the reference uses
    ("bb72_custom_r6_p0.003", 600, 6, "lsd_cs", 5),
    ("bb72_custom_r6_p0.003", 600, 6, "lsd_e", 4),
    ("hgp225_cardinal_r3_p0.01", 150, 10, "lsd_cs", 2),
    ("bb144_custom_r12_p0.003", 400, 30, "lsd_cs", 1),
    ("bb144_custom_r12_p0.003", 200, 20, "lsd_e", 6),
])
def test_bplsd_higher_order_bit_exact(gpu, name, shots, max_iter, method, order):
    """lsd_order > 0 (reference bplsd.py:10,54 forward lsd_method / lsd_order; its own calls pass lsd_order = 1:
    /root/reference/tests/test_decoders.py:136): growth stage + per-cluster sweep on the device against the oracle's oq_lsd,
    bit for bit, and the sweep really replaces LSD-0 solutions."""
    H, L, pri = helpers.dem_matrices(name)
    synd, _, _ = orc.sample_dem(H, L, pri, seed=14, shot0=0, B=shots)
    err, status, dec = _gpu_decode(H, pri, synd, max_iter, osd=method, order=order)
    g, prm = _oracle(H, pri, max_iter, method, order=order)
    ref, flags = g.decode_batch(synd, prm)
    used = (status >> 17) & 1
    assert np.array_equal(used, 1 - flags[:, 0]) and used.sum() > 10
    assert np.array_equal((status >> 20) & 0xFFF, np.minimum(flags[:, 2], 4095)), "pivot counts differ"
    bad = np.flatnonzero((err != ref).any(axis=1))
    assert bad.size == 0, "LSD-w output differs on shots %s" % bad[:10]
    Hd = np.asarray(H.todense(), dtype=np.int64)
    assert np.array_equal((err.astype(np.int64) @ Hd.T) % 2, synd)
    err0, _, _ = _gpu_decode(H, pri, synd, max_iter, osd="lsd_0")
    assert (err != err0).any(axis=1).sum() > 3, "the higher order never changed a solution: test does not exercise the sweep"


def test_bplsd_higher_order_crafted_soft_information(gpu):
    """LSD-w alone on soft information that forces ties, big merged clusters with many non-pivot faults (more candidate
    positions than the 64 whose images are kept; exhaustive patterns on the first 10), and inconsistent syndromes."""
    import torch
    from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits
    H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
    m, n = H.shape
    rng = np.random.default_rng(18)
    B = 20
    synd, _, _ = orc.sample_dem(H, L, pri, seed=6, shot0=0, B=B)
    synd[B - 3:] = (rng.random((3, m)) < 0.2).astype(np.uint8)            # mostly inconsistent (rank 1002 < 1008)
    llr = rng.normal(size=(B, n)).astype(np.float32)
    llr[:5] = 2.5
    llr[5:10] = rng.choice(np.array([-1.0, 0.25, 3.0], np.float32), size=(5, n))
    wg = WindowGraph(H, pri)
    g = orc.Graph(H, pri)
    for method, order in (("lsd_cs", 1), ("lsd_cs", 12), ("lsd_cs", 64), ("lsd_e", 10)):
        dec = BatchDecoder(wg, max_iter=1, osd_method=method, osd_order=order)
        bits, status = dec.osd0(torch.from_numpy(synd).cuda(), torch.from_numpy(llr).cuda())
        err, status = unpack_bits(bits, n).cpu().numpy(), status.cpu().numpy()
        swept = 0
        for b in range(B):
            ref, st = g.lsd(synd[b], llr[b].astype(np.float64), method, order, fixed=True)
            assert np.array_equal(err[b], ref), (method, order, b, st)
            assert ((status[b] >> 20) & 0xFFF) == min(st["pivots"], 4095) and bool(status[b] & (1 << 18)) == st["inconsistent"], (b, st)
            swept += st["replaced"]
        assert swept > 0, (method, order)


def test_reference_decoder_tests_bposd_and_bplsd_phenom(gpu):
    """/root/reference/tests/test_decoders.py:88-159 on the device: BPC code (lift 15, factor 3), cardinal circuit seed 1,
    p = 5e-4, 10 rounds, W = 5, F = 3, eff_error_rate_per_fault = p (depth + 3), max_iter = 10, osd_order = 1 / lsd_order = 1,
    wrapper defaults otherwise -- with the reference's own acceptance thresholds, on 20 000 shots instead of 50, and bit for
    bit against the oracle on the first 96."""
    from quits_amd.decoder import sliding_window_bplsd_phenom_mem, sliding_window_bposd_phenom_mem
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.decoder.device import DemSampler
    from quits_amd.decoder.sliding_window import phenom_window_set
    from quits_amd.dem import Circuit
    name = "bpc_cardinal_r10_p0.0005"
    meta = helpers.circuit_index()[name]
    cd = helpers.code("bpc_15_3")
    hz, lz = cd["hz"], cd["lz"]
    nz = hz.shape[0]
    R, W, F, p = 10, 5, 3, 5e-4
    eff = p * (meta["depth"] + 3)
    H, L, pri = detector_error_model_to_matrix(Circuit(helpers.circuit_text(name)).detector_error_model())
    det, obs = DemSampler(H, L, pri).sample(20000, seed=1)
    obs = obs.cpu().numpy()
    for fn, kw, lim_pl, lim_lfr, method in ((sliding_window_bposd_phenom_mem, dict(max_iter=10, osd_order=1), 0.25, 0.08, "osd_cs"),
                                            (sliding_window_bplsd_phenom_mem, dict(max_iter=10, lsd_order=1), 0.3, 0.1, "lsd_cs")):
        pred = fn(det, hz, lz, W, F, eff_error_rate_per_fault=eff, tqdm_on=False, **kw)
        pL = np.mean((obs - pred).any(axis=1))                       # the reference's formulas (test_decoders.py:68-69)
        lfr = 1 - (1 - pL) ** (1 / R)
        assert pL <= lim_pl and lfr <= lim_lfr, (fn.__name__, pL, lfr)
        a, b, pp, d = phenom_window_set(hz, lz, W, F, R, eff, eff)
        wins = [{"H": a[k], "L": b[k], "priors": pp[k], "U": d[k] if k < len(d) else None, "row0": F * k * nz} for k in range(len(a))]
        ref, stats = orc.sliding_window_decode(wins, nz, det[:96].cpu().numpy(),
                                               orc.make_params("product_sum", "serial", 10, method, 1, 1.0, orc.FORM_LDPC_F32))
        assert np.array_equal(pred[:96], ref.astype(np.int64)), fn.__name__


def test_bplsd_crafted_soft_information_and_inconsistent(gpu):
    """LSD alone (qd_osd0_batch on a BP-LSD decoder) on soft information that forces ties (flat LLRs: growth by index), big
    merges (random LLRs) and syndromes outside the column space (flagged, no hang)."""
    import torch
    from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits
    H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
    m, n = H.shape
    rng = np.random.default_rng(8)
    B = 24
    synd, _, _ = orc.sample_dem(H, L, pri, seed=3, shot0=0, B=B)
    synd[B - 4:] = (rng.random((4, m)) < 0.2).astype(np.uint8)            # arbitrary: mostly inconsistent (rank 1002 < 1008)
    llr = rng.normal(size=(B, n)).astype(np.float32)
    llr[:6] = 2.5
    llr[6:12] = rng.choice(np.array([-1.0, 0.25, 3.0], np.float32), size=(6, n))
    wg = WindowGraph(H, pri)
    dec = BatchDecoder(wg, max_iter=1, osd_method="lsd_0")
    bits, status = dec.osd0(torch.from_numpy(synd).cuda(), torch.from_numpy(llr).cuda())
    err, status = unpack_bits(bits, n).cpu().numpy(), status.cpu().numpy()
    g = orc.Graph(H, pri)
    for b in range(B):
        ref, st = g.lsd0(synd[b], llr[b].astype(np.float64))
        assert np.array_equal(err[b], ref), b
        assert ((status[b] >> 20) & 0xFFF) == min(st["pivots"], 4095) and bool(status[b] & (1 << 18)) == st["inconsistent"], (b, st)
    assert (status[B - 4:] & (1 << 18)).any()


@pytest.mark.parametrize("rows", [1296, 1708])
def test_bplsd_wide_windows(gpu, rows):
    """The LSD kernel's other row-block instantiations (24 and 32 checks per lane: 1025..1536 and 1537..2048 checks), LSD alone
    on crafted soft information over block-diagonal matrices assembled from the committed windows; vs the oracle, bit for bit."""
    import torch
    from scipy.sparse import block_diag, csc_matrix
    from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits
    Ha, La, pa = helpers.dem_matrices("bb144_custom_r12_p0.003")
    if rows == 1296:
        Hb, Lb, pb = helpers.dem_matrices("bb72_custom_r6_p0.003")
    else:
        Hb, pb = csc_matrix(Ha[:700]), pa
    H = csc_matrix(block_diag([Ha, Hb], format="csc"))
    pri = np.concatenate([pa, pb])
    m, n = H.shape
    assert m == rows and n <= 65535
    rng = np.random.default_rng(rows)
    B = 12
    e = (rng.random((B, n)) < np.minimum(4 * pri, 0.4)).astype(np.uint8)
    synd = np.ascontiguousarray(np.asarray((csc_matrix(e) @ H.T).todense()) % 2, dtype=np.uint8)
    llr = rng.normal(size=(B, n)).astype(np.float32) + 2.0
    llr[:3] = np.log((1 - pri) / pri).astype(np.float32)
    llr[e.astype(bool)] -= 3.0
    wg = WindowGraph(H, pri)
    dec = BatchDecoder(wg, max_iter=1, osd_method="lsd_0")
    bits, status = dec.osd0(torch.from_numpy(synd).cuda(), torch.from_numpy(llr).cuda())
    err, status = unpack_bits(bits, n).cpu().numpy(), status.cpu().numpy()
    g = orc.Graph(H, pri)
    Hd = np.asarray(H.todense(), dtype=np.int64)
    big = 0
    for b in range(B):
        ref, st = g.lsd0(synd[b], llr[b].astype(np.float64))
        assert np.array_equal(err[b], ref), b
        assert ((status[b] >> 20) & 0xFFF) == min(st["pivots"], 4095) and not st["inconsistent"], (b, st)
        big = max(big, st["pivots"])
    assert np.array_equal((err.astype(np.int64) @ Hd.T) % 2, synd)
    assert big > 128, "no shot reached the Q planes kept in HBM"


@pytest.mark.parametrize("rows", [1296, 1708, 2016])
def test_osd0_many_pivot_kernel_wide_windows(gpu, rows):
    """qd_osd0_sr_kernel's instantiations with three and four rows per thread (1025..1536 and 1537..2048 checks), OSD-0 alone on
    block-diagonal matrices assembled from the committed windows: early stops inside a batch (the drain and the sequential pivot count),
    shots that run deep into the Q planes kept in L2 (more than 256 pivots), ties; vs the oracle, bit for bit."""
    from scipy.sparse import block_diag, csc_matrix
    Ha, La, pa = helpers.dem_matrices("bb144_custom_r12_p0.003")
    if rows == 1296:
        Hb, Lb, pb = helpers.dem_matrices("bb72_custom_r6_p0.003")
    elif rows == 1708:
        Hb, pb = csc_matrix(Ha[:700]), pa
    else:
        Hb, pb = Ha, pa
    H = csc_matrix(block_diag([Ha, Hb], format="csc"))
    pri = np.concatenate([pa, pb])
    m, n = H.shape
    assert m == rows and n <= 49152
    rng = np.random.default_rng(rows)
    B = 16
    e = (rng.random((B, n)) < np.minimum(4 * pri, 0.4)).astype(np.uint8)
    e[B - 4:] = (rng.random((4, n)) < np.minimum(40 * pri, 0.5)).astype(np.uint8)       # heavy errors: hundreds of pivots
    synd = np.ascontiguousarray(np.asarray((csc_matrix(e) @ H.T).todense()) % 2, dtype=np.uint8)
    llr = rng.normal(size=(B, n)).astype(np.float32) + 2.0
    llr[:3] = np.log((1 - pri) / pri).astype(np.float32)
    llr[e.astype(bool)] -= 3.0
    llr[3] = 1.0                                                                           # one key for every fault: order by index
    err, status = _osd_only(H, pri, synd, llr)
    g = orc.Graph(H, pri)
    Hd = np.asarray(H.todense(), dtype=np.int64)
    big = 0
    for b in range(B):
        ref, st = g.osd0(synd[b], llr[b].astype(np.float64), stop_early=True)
        assert np.array_equal(err[b], ref), b
        assert ((status[b] >> 20) & 0xFFF) == min(st["pivots"], 4095) and not st["inconsistent"], (b, st)
        big = max(big, st["pivots"])
    assert np.array_equal((err.astype(np.int64) @ Hd.T) % 2, synd)
    assert big > 256, "no shot reached the Q planes kept in L2"


def test_bplsd_sliding_window_functions(gpu):
    """sliding_window_bplsd_circuit_mem / _phenom_mem (reference bplsd.py:10,54) on the device against the oracle's loop."""
    from quits_amd.decoder import BpLsdDecoder, sliding_window_bplsd_circuit_mem, sliding_window_bplsd_phenom_mem
    from quits_amd.decoder.base import spacetime, window_count
    from quits_amd.dem import Circuit
    name = "bb72_custom_r6_p0.003"
    cd = helpers.code("bb72")
    hz, lz = cd["hz"], cd["lz"]
    nz = hz.shape[0]
    H, L, pri = helpers.dem_matrices(name)
    det, obs, _ = orc.sample_dem(H, L, pri, seed=4, shot0=0, B=300)
    circ = Circuit(helpers.circuit_text(name))
    pred = sliding_window_bplsd_circuit_mem(det, circ, hz, lz, 3, 1, max_iter=10, lsd_order=0, bp_method="minimum_sum",
                                            schedule="parallel", lsd_method="lsd_cs")
    ncr, _, _ = window_count(6, 3, 1)
    checks, commits, priors, updates = spacetime(circ, hz, 3, 1, ncr)
    wins = [{"H": checks[k], "L": commits[k], "priors": priors[k], "U": updates[k] if k < ncr else None, "row0": k * nz}
            for k in range(len(checks))]
    ref, stats = orc.sliding_window_decode(wins, nz, det, orc.make_params("minimum_sum", "parallel", 10, "lsd_0", 0, 1.0, orc.FORM_LDPC_F64),
                                           device_grid=True)
    assert stats["osd_calls"] > 50 and pred.dtype == np.int64 and np.array_equal(pred, ref.astype(np.int64))
    assert (pred != obs).any(axis=1).mean() < 0.2
    # the reference wrapper's own defaults (product_sum, serial, max_iter 2, lsd_cs order 0): the general BP kernel + LSD-0
    pred2 = sliding_window_bplsd_circuit_mem(det[:96], circ, hz, lz, 3, 1)
    ref2, _ = orc.sliding_window_decode(wins, nz, det[:96], orc.make_params("product_sum", "serial", 2, "lsd_0", 0, 1.0, orc.FORM_LDPC_F32))
    assert np.array_equal(pred2, ref2.astype(np.int64))
    # phenomenological variant runs and returns the right shape; plug-in class per shot
    pp = sliding_window_bplsd_phenom_mem(det[:64], hz, lz, 3, 1, eff_error_rate_per_fault=0.03, max_iter=10,
                                         bp_method="minimum_sum", schedule="parallel")
    assert pp.shape == (64, lz.shape[0]) and pp.dtype == np.int64
    d1 = BpLsdDecoder(checks[1], channel_probs=priors[1], max_iter=10, bp_method="minimum_sum", schedule="parallel", lsd_method="lsd_cs", lsd_order=0)
    go, prm = _oracle(checks[1], priors[1], 10, "lsd_0")
    s1 = det[7, nz:4 * nz]
    assert np.array_equal(d1.decode(s1.astype(int)), go.decode_batch(s1.reshape(1, -1), prm)[0][0])


def test_osdw_panel_kernel_and_row_form_agree(gpu, monkeypatch):
    """Higher-order OSD has two full-rank eliminations: the panel kernel (osd_cs.hip, qd_osdcs_kernel: the default wherever its layout
    takes the window) and the elimination by row (qd_osd0_reg_kernel<.., true>: QD_OSDCS_OLD=1, and every window of more than 1408
    checks).  Same pivots, same sweep, same bits -- on the headline window both ways, and on a 1708-check block-diagonal window (row
    form only) vs the oracle."""
    import torch
    from scipy.sparse import block_diag, csc_matrix
    from quits_amd.decoder.device import BatchDecoder, WindowGraph, unpack_bits
    Ha, La, pa = helpers.dem_matrices("bb144_custom_r12_p0.003")
    for H, pri, shots in ((Ha, pa, 8), (csc_matrix(block_diag([Ha, csc_matrix(Ha[:700])], format="csc")), np.concatenate([pa, pa]), 4)):
        m, n = H.shape
        rng = np.random.default_rng(m)
        e = (rng.random((shots, n)) < 2 * pri).astype(np.uint8)
        synd = np.ascontiguousarray(np.asarray((csc_matrix(e) @ H.T).todense()) % 2, dtype=np.uint8)
        llr = (np.log((1 - pri) / pri)[None, :] + rng.normal(size=(shots, n))).astype(np.float32)
        llr[e.astype(bool)] -= 4.0
        wg = WindowGraph(H, pri)
        out = []
        for rows in ("0", "1"):
            monkeypatch.setenv("QD_OSDCS_OLD", rows)
            dec = BatchDecoder(wg, max_iter=1, osd_method="osd_cs", osd_order=3)
            assert dec.info()["post_kernel"] == ("qd_osdcs_kernel" if (rows == "0" and m <= 1408) else "qd_osd0_reg_kernel<row form>")
            bits, status = dec.osd0(torch.from_numpy(synd).cuda(), torch.from_numpy(llr).cuda())
            out.append((unpack_bits(bits, n).cpu().numpy(), status.cpu().numpy()))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        g = orc.Graph(H, pri)
        for b in range(shots):
            ref, st = g.osd_w(synd[b], llr[b].astype(np.float64), "osd_cs", 3, fixed=True)
            assert np.array_equal(out[0][0][b], ref), (m, b, st)


@pytest.mark.parametrize("scale,shots,max_iter", [(1.0 / 3.0, 1024, 30), (1.0, 256, 12)])
def test_wide_scatter_kernel_and_gather_kernel_agree(gpu, monkeypatch, scale, shots, max_iter):
    """configs[4] windows (QLP [[1020,136]], W = 3: 1350 checks of up to 78 faults, 18 900 faults) run flooding min-sum in the
    several-checks-per-lane scatter kernel (bp_scatter_wide.hip: 512 lanes x 3 checks, or 704 x 2; three sign words).  Same contract as the one-check-per-lane
    form: hard decisions, status words and OSD-0 outputs identical to the gather kernel's (QD_NO_SCATTER=1), also through the recheck
    pass, and equal to the double-precision oracle on the same LLR grid.  Priors of the fixture (p = 3e-3, above threshold: nearly
    every shot runs all iterations) and scaled to ~1e-3 (most shots converge, at different iterations)."""
    from scipy.sparse import csc_matrix
    from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph, unpack_bits
    w = helpers.window_set("qlp1020_cardinal_r20_p0.003", 3, 1)[1]
    H, pri = w["H"], np.asarray(w["priors"], dtype=np.float64) * scale
    assert H.shape == (1350, 18900)
    L = csc_matrix(np.ones((1, H.shape[1]), dtype=np.uint8))      # the sampler wants an observable matrix over the window's faults; not used
    det, _ = DemSampler(H, L, pri).sample(shots, seed=5, shot0=1)
    det[0] = 0
    wg = WindowGraph(H, pri)
    out = {}
    for tag, env in (("gather", {"QD_NO_SCATTER": "1"}), ("scatter", {}), ("recheck", {"QD_SCATTER_M2_LIMIT": "30000"}),
                     ("two_checks_704_lanes", {"QD_SCATTER_WIDE_T704": "1"})):
        for k in ("QD_NO_SCATTER", "QD_SCATTER_M2_LIMIT", "QD_SCATTER_WIDE_T704"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dec = BatchDecoder(WindowGraph(H, pri), max_iter=max_iter, osd_method="osd_0")
        assert dec.info()["scatter_kernel"] == (tag != "gather"), (tag, dec.info())
        assert dec.info()["scatter_wide_kernel"] == (tag != "gather"), (tag, dec.info())
        for stage in (1, 3):
            bits, status = dec.decode(det, stage=stage)
            out[(tag, stage)] = (unpack_bits(bits, wg.n).cpu().numpy(), status.cpu().numpy())
    for tag in ("scatter", "recheck", "two_checks_704_lanes"):
        for stage in (1, 3):
            assert np.array_equal(out[(tag, stage)][1], out[("gather", stage)][1]), (tag, stage)
            assert np.array_equal(out[(tag, stage)][0], out[("gather", stage)][0]), (tag, stage)
    st = out[("scatter", 1)][1]
    conv = ((st >> 16) & 1).mean()
    assert (0.05 < conv < 0.95) if scale < 1.0 else (conv < 0.9), conv       # a mix of iteration counts / mostly max_iter
    nref = 48
    g, prm = _oracle(H, pri, max_iter, "osd_0")
    ref, flags = g.decode_batch(np.ascontiguousarray(det[:nref].cpu().numpy()), prm)
    assert np.array_equal(out[("scatter", 3)][0][:nref], ref)


@pytest.mark.parametrize("name,shots,max_iter", [
    ("bb144_custom_r12_p0.003", 4096, 50),        # the headline window: 1008 checks on 1024 lanes, rows of 16..35 faults (two sign words)
    ("bb72_custom_r6_p0.003", 4096, 30),
    ("hgp225_cardinal_r3_p0.01", 2048, 20),       # rows of very different weights in one wavefront (the predicated tail groups)
])
def test_scatter_kernel_and_gather_kernel_agree(gpu, monkeypatch, name, shots, max_iter):
    """Flooding min-sum on the LLR grid has two kernels: the scatter form (checks add their messages to integer accumulators; the
    default where the window fits it: bp_scatter_wide.hip with two checks per lane on half the lanes, or bp_scatter.hip with one,
    QD_SCATTER_CPL1=1) and the gather form (bp_kernels.hip, QD_NO_SCATTER=1).  All are exact, so
    hard decisions, status words (iteration counts, convergence) and the OSD outputs computed from the posteriors must be
    identical -- also when the scatter kernel's cheap exactness bound is made to fail for most shots (QD_SCATTER_M2_LIMIT), so that
    they are decoded again by the gather kernel in the recheck pass -- and equal to the double-precision oracle."""
    import torch
    from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph, unpack_bits
    H, L, pri = helpers.dem_matrices(name)
    det, _ = DemSampler(H, L, pri).sample(shots, seed=11, shot0=3)
    det[0] = 0
    wg = WindowGraph(H, pri)
    out = {}
    for tag, env in (("gather", {"QD_NO_SCATTER": "1"}), ("scatter", {}), ("recheck", {"QD_SCATTER_M2_LIMIT": "40000"}),
                     ("one_check_per_lane", {"QD_SCATTER_CPL1": "1"}), ("natural_rounds", {"QD_SCATTER_NATURAL_ROUNDS": "1"})):
        for k in ("QD_NO_SCATTER", "QD_SCATTER_M2_LIMIT", "QD_SCATTER_CPL1", "QD_SCATTER_NATURAL_ROUNDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        wg_t = WindowGraph(H, pri)       # the kernel shape is a property of the graph object (qd_graph_create)
        dec = BatchDecoder(wg_t, max_iter=max_iter, osd_method="osd_0")
        assert dec.info()["scatter_kernel"] == (tag != "gather"), (tag, dec.info())
        assert dec.info()["scatter_wide_kernel"] == (tag not in ("gather", "one_check_per_lane")), (tag, dec.info())
        for stage in (1, 3):
            bits, status = dec.decode(det, stage=stage)
            out[(tag, stage)] = (unpack_bits(bits, wg.n).cpu().numpy(), status.cpu().numpy())
    for tag in ("scatter", "recheck", "one_check_per_lane", "natural_rounds"):
        for stage in (1, 3):
            assert np.array_equal(out[(tag, stage)][0], out[("gather", stage)][0]), (tag, stage)
            assert np.array_equal(out[(tag, stage)][1], out[("gather", stage)][1]), (tag, stage)
    nref = 256
    g, prm = _oracle(H, pri, max_iter, "osd_0")
    ref, flags = g.decode_batch(np.ascontiguousarray(det[:nref].cpu().numpy()), prm)
    assert np.array_equal(out[("scatter", 3)][0][:nref], ref)


@pytest.mark.parametrize("fixture,window", [("bb144_custom_r12_p0.003", None), ("bb144_custom_r12_p0.003", (3, 1, 1)), ("bb72_custom_r6_p0.003", None)])
def test_scatter_accumulator_banks_and_walk_do_not_change_results(gpu, monkeypatch, fixture, window):
    """The scatter kernels keep their accumulators in a slot order of their own: a bank per fault balanced over the groups of 32 checks
    (scatter_banks) and a walk per check found by matching (scatter_walk; qd_api.hip).  Neither may change a bit: the same shots through
    the round-3 layout (QD_SCATTER_BANKS_BY_SLOT=1: degree-sorted bit slots), through the greedy walk (QD_SCATTER_WALK_GREEDY=1) and through the
    default give identical decisions, status words and posteriors-driven OSD-0 outputs; and the default's modelled LDS cycles per pass are within
    15 % of the conflict-free count."""
    import torch
    from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
    if window is None:
        H, L, pri = helpers.dem_matrices(fixture)
    else:
        w = helpers.window_set(fixture, window[0], window[1])[window[2]]
        H, pri = w["H"], w["priors"]
        L = H[:8]
    det, _ = DemSampler(H, L, pri).sample(1024, seed=11)
    outs, infos = [], []
    for env in ({}, {"QD_SCATTER_BANKS_BY_SLOT": "1"}, {"QD_SCATTER_WALK_GREEDY": "1"}, {"QD_SCATTER_BANKS_BY_SLOT": "1", "QD_SCATTER_WALK_GREEDY": "1"}):
        for k in ("QD_SCATTER_BANKS_BY_SLOT", "QD_SCATTER_WALK_GREEDY"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = WindowGraph(H, pri)
        d = BatchDecoder(g, max_iter=50, osd_method="osd_0")
        assert d.info()["scatter_kernel"]
        bits, st = d.decode(det)
        torch.cuda.synchronize()
        outs.append((bits.cpu().numpy(), st.cpu().numpy()))
        infos.append(g.info())
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
    assert 0 < infos[0]["scatter_walk_ideal"] <= infos[0]["scatter_walk_cycles"] <= 1.15 * infos[0]["scatter_walk_ideal"], infos[0]
    assert infos[3]["scatter_walk_cycles"] > infos[0]["scatter_walk_cycles"]          # the old layout and walk cost more LDS cycles in the same model
