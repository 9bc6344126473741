"""The CPU oracle (oracle/qd_oracle.c) against the reference's golden vectors and its own invariants."""
import os

import numpy as np
import pytest
from scipy.sparse import csc_matrix, csr_matrix

import helpers
import oracle as orc


def _gf2_gauss_solve(A, b, order):
    """Dense textbook elimination in a given column order (independent of the oracle's bookkeeping):
    returns (solution using the first independent columns, rank, consistent)."""
    A = A.copy() % 2
    b = b.copy() % 2
    m, n = A.shape
    piv_rows, piv_cols = [], []
    used = np.zeros(m, bool)
    for c in order:
        rows = np.flatnonzero((A[:, c] == 1) & ~used)
        if rows.size == 0:
            continue
        p = rows[0]
        used[p] = True
        for r in np.flatnonzero(A[:, c]):
            if r != p:
                A[r] ^= A[p]
                b[r] ^= b[p]
        piv_rows.append(p)
        piv_cols.append(c)
    x = np.zeros(n, np.uint8)
    for p, c in zip(piv_rows, piv_cols):
        x[c] = b[p]
    return x, len(piv_cols), not b[~used].any()


def test_gf2_known_answers_from_reference():
    """G6: rank and solvability as computed by the reference's gf2_util (gf2_rank / gf2_solve)."""
    z = np.load(helpers.GOLD + "/gf2.npz")
    rng = np.random.default_rng(0)
    for i in range(int(z["count"][0])):
        m, n = (int(v) for v in z["shape%d" % i])
        A = np.unpackbits(z["A%d" % i], axis=1)[:, :n]
        if not A.any(axis=0).all() or not A.any(axis=1).all():
            cols = np.flatnonzero(A.any(axis=0))
        g = None
        # the oracle needs priors in (0,1) and column weights <= 64
        if A.sum(axis=0).max() > 64:
            continue
        keep = np.flatnonzero(A.any(axis=0))
        g = orc.Graph(csr_matrix(A), np.full(n, 0.1))
        assert g.rank() == int(z["rank%d" % i][0])
        b = z["b%d" % i]
        llr = rng.normal(size=n)
        e, st = g.osd0(b, llr, stop_early=True)
        assert not st["inconsistent"] and np.array_equal(A @ e % 2, b)
        e_full, _ = g.osd0(b, llr, stop_early=False)
        assert np.array_equal(e, e_full)
        x, rank, ok = _gf2_gauss_solve(A, b, list(orc.column_order(llr)))
        assert ok and rank == g.rank() and np.array_equal(x, e)
        if int(z["bbad_ok%d" % i][0]) == 0:            # a right-hand side the reference reports unsolvable
            _, st_bad = g.osd0(z["bbad%d" % i], llr, stop_early=True)
            assert st_bad["inconsistent"]


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    out = np.zeros(4, np.uint32)
    L = orc.lib()
    L.oq_philox(0, 0, 0, 0, 0, 0, out)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    L.oq_philox(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, out)
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    L.oq_philox(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0, out)
    assert [hex(v) for v in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    assert L.oq_prob_threshold(0.5) == 2 ** 31 and L.oq_prob_threshold(0.0) == 0


def test_sampler_statistics_and_determinism():
    H, Lm, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    s1, o1, nf = orc.sample_dem(H, Lm, pri, seed=3, shot0=0, B=4000)
    s2, o2, _ = orc.sample_dem(H, Lm, pri, seed=3, shot0=1000, B=500)
    assert np.array_equal(s1[1000:1500], s2) and np.array_equal(o1[1000:1500], o2)      # counter-based: shots are addressable
    assert abs(nf.mean() - pri.sum()) < 4 * np.sqrt(pri.sum() / 4000)
    s3, _, _ = orc.sample_dem(H, Lm, pri, seed=4, shot0=0, B=100)
    assert not np.array_equal(s1[:100], s3)


@pytest.mark.parametrize("form", [orc.FORM_LDPC_F64, orc.FORM_COMPRESSED_F32, orc.FORM_COMPRESSED_F64, orc.FORM_LDPC_F32])
def test_bp_invariants(form):
    H, Lm, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    g = orc.Graph(H, pri)
    prm = orc.make_params("minimum_sum", "parallel", 30, "osd_0", 0, 1.0, form)
    conv, dec, llr, it = g.bp(np.zeros(H.shape[0], np.uint8), prm)
    assert conv and it == 0 and not dec.any()                       # zero syndrome -> zero correction
    synd, obs, _ = orc.sample_dem(H, Lm, pri, seed=9, shot0=0, B=200)
    Hd = np.asarray(H.todense(), dtype=np.int64)
    nconv = 0
    for i in range(200):
        conv, dec, llr, it = g.bp(synd[i], prm)
        if conv:
            nconv += 1
            assert np.array_equal(Hd @ dec % 2, synd[i]) and 1 <= it <= 30
            assert np.array_equal(dec, (llr <= 0).astype(np.uint8))
        else:
            assert it == 30
    assert nconv > 100
    # single faults: BP must converge to a correction with the fault's observable signature
    Ld = np.asarray(Lm.todense(), dtype=np.int64)
    for j in (0, 17, 400, 1300, 2591):
        s = Hd[:, j] % 2
        conv, dec, _, _ = g.bp(s.astype(np.uint8), prm)
        assert conv and np.array_equal(Ld @ dec % 2, Ld[:, j] % 2)


def test_forms_agree_statistically_and_schedules_run():
    H, Lm, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    g = orc.Graph(H, pri)
    synd, obs, _ = orc.sample_dem(H, Lm, pri, seed=10, shot0=0, B=400)
    res = {}
    for name, prm in (("ldpc", orc.make_params("minimum_sum", "parallel", 30, "osd_0", 0, 1.0, orc.FORM_LDPC_F64)),
                      ("gpu", orc.make_params("minimum_sum", "parallel", 30, "osd_0", 0, 1.0, orc.FORM_COMPRESSED_F32)),
                      ("c64", orc.make_params("minimum_sum", "parallel", 30, "osd_0", 0, 1.0, orc.FORM_COMPRESSED_F64)),
                      ("ps", orc.make_params("product_sum", "parallel", 30, "osd_0", 0, 1.0, orc.FORM_LDPC_F64)),
                      ("serial", orc.make_params("minimum_sum", "serial", 10, "osd_0", 0, 1.0, orc.FORM_LDPC_F64)),
                      ("serial_ps", orc.make_params("product_sum", "serial", 10, "osd_cs", 2, 1.0, orc.FORM_LDPC_F64))):
        err, flags = g.decode_batch(synd, prm)
        assert np.array_equal((err.astype(np.int64) @ np.asarray(H.todense(), dtype=np.int64).T) % 2, synd), name
        res[name] = (np.asarray(Lm @ err.T % 2).T != obs).any(axis=1).mean()
    # same algorithm, different rounding: logical error rates within Monte-Carlo noise of each other
    assert abs(res["ldpc"] - res["gpu"]) < 0.06 and abs(res["ldpc"] - res["c64"]) < 0.06
    assert all(0.0 <= v < 0.35 for v in res.values())
    with pytest.raises(ValueError):      # the compressed form only exists for flooding min-sum
        g.decode_batch(synd[:2], orc.make_params("product_sum", "parallel", 5, "osd_0", 0, 1.0, orc.FORM_COMPRESSED_F32))


def test_osd_cs_never_worse_than_osd0():
    H, Lm, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    g = orc.Graph(H, pri)
    synd, _, _ = orc.sample_dem(H, Lm, pri, seed=12, shot0=0, B=60)
    prm = orc.make_params("minimum_sum", "parallel", 8, "osd_0", 0, 1.0, orc.FORM_LDPC_F64)
    w = np.log(1.0 / pri)
    checked = 0
    for i in range(60):
        conv, dec, llr, _ = g.bp(synd[i], prm)
        if conv:
            continue
        e0, _ = g.osd0(synd[i], llr)
        e1 = g.osd_w(synd[i], llr, "osd_cs", 1)
        ee = g.osd_w(synd[i], llr, "osd_e", 3)
        Hd = np.asarray(H.todense(), dtype=np.int64)
        for e in (e0, e1, ee):
            assert np.array_equal(Hd @ e % 2, synd[i])
        assert w @ e1 <= w @ e0 + 1e-9 and w @ ee <= w @ e0 + 1e-9
        checked += 1
        if checked == 6:
            break
    assert checked >= 3


def test_oracle_plugin_surface():
    H, Lm, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    d = orc.OracleBpOsdDecoder(csc_matrix(H), channel_probs=pri, max_iter=10, bp_method="minimum_sum",
                               schedule="parallel", osd_method="osd_0", osd_order=0)
    s, _, _ = orc.sample_dem(H, Lm, pri, seed=1, shot0=5, B=1)
    e = d.decode(s[0].astype(int))
    assert e.shape == (H.shape[1],) and np.array_equal(np.asarray(H @ e % 2).ravel(), s[0])
    with pytest.raises(ValueError):
        orc.OracleBpOsdDecoder(csc_matrix(H))


def test_oracle_float_functions_equal_the_products_bit_for_bit(tmp_path):
    """oracle/oq_math.h (the checker's own e^-|x|, (a + b) / (1 + a b) and -log(u)) and quits_amd/csrc/qd_math.h (the product's,
    here compiled for the host) share no code and must return the same bits: 400 000 inputs incl. the range limits."""
    import ctypes
    import subprocess
    src = tmp_path / "qd_math_host.c"
    src.write_text('#include "qd_math.h"\n#include <stdint.h>\n'
                   'void f(int kind, const float *x, const float *x2, float *y, int64_t n) { for (int64_t i = 0; i < n; i++) '
                   'y[i] = kind == 0 ? qd_exp_neg(x[i]) : (kind == 1 ? qd_neg_log(x[i]) : qd_ucomb(x[i], x2[i])); }\n')
    so = tmp_path / "qd_math_host.so"
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-std=c11",
                           "-I", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quits_amd", "csrc"),
                           "-o", str(so), str(src), "-lm"])
    lib = ctypes.CDLL(str(so))
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    lib.f.argtypes = [ctypes.c_int, f32p, f32p, f32p, ctypes.c_int64]
    rng = np.random.default_rng(11)
    edge = np.float32([0.0, -0.0, 0.5, -0.5, 0.34657359, 0.3465736, 87.0, 87.00001, 1e4, -1e4, 1e-30, 1.0, 0.70710677, 0.7071068,
                       1.17549435e-38, 1e-45, 0.99999994, np.inf, -np.inf])
    x = np.concatenate([rng.uniform(-95, 95, 150000), rng.uniform(-1, 1, 150000), rng.normal(0, 1e-3, 50000),
                        rng.uniform(-1, 1, 50000) * 10.0 ** rng.uniform(-30, 0, 50000), edge]).astype(np.float32)
    u = np.concatenate([rng.uniform(0, 1, 200000), np.exp(-rng.uniform(0, 95, 150000)), 1 - 10.0 ** rng.uniform(-8, -1, 50000),
                        np.abs(edge[np.isfinite(edge)]).clip(0, 1)]).astype(np.float32)
    u2 = rng.permutation(u)
    for kind, name, a1, a2 in ((0, "exp_neg", x, x), (1, "neg_log", u, u), (2, "ucomb", u, u2)):
        y = np.empty_like(a1)
        lib.f(kind, a1, a2, y, a1.size)
        assert np.array_equal(y.view(np.uint32), orc.math_f32(name, a1, a2).view(np.uint32)), name


def test_float_elementary_functions_against_libm():
    """e^-|x| (sign of x kept), (a + b) / (1 + a b) and -log(u) as the float product-sum forms evaluate them (the oracle's oq_math.h;
    bit-identical to the product's qd_math.h by the test above) against double precision: a few ulp over the whole float range the
    device uses (|x| <= 87, u >= 2^-126), and the identities the check update rests on:
    tanh(x1/2) tanh(x2/2) has u = C(u1, u2), and log((1 + t) / (1 - t)) = -log(u)."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-87, 87, 100000), rng.uniform(-1, 1, 100000), rng.normal(0, 1e-3, 20000),
                        [0.0, -0.0, 0.5, -0.5, 17.0, 19.0, 35.0, 50.0, 86.9, 1e-30]]).astype(np.float32)
    y = orc.math_f32("exp_neg", x)
    ref = np.exp(-np.abs(x.astype(np.float64)))
    assert (np.abs(np.abs(y) - ref) / ref).max() < 4 * 2.0 ** -24
    assert np.array_equal(np.signbit(y), np.signbit(x)) and np.all(np.abs(y) <= 1)
    assert abs(float(orc.math_f32("exp_neg", np.float32([1e4]))[0]) - np.exp(-87.0)) < 1e-44       # saturates at |x| = 87
    u = np.concatenate([rng.uniform(0, 1, 100000), np.exp(-rng.uniform(0, 87, 100000)), 1 - 10.0 ** rng.uniform(-7.2, -1, 50000),
                        [1.0, 0.5, 0.70710677, 0.7071068, 1.17549435e-38]]).astype(np.float32)
    y = orc.math_f32("neg_log", u)
    ref = -np.log(u.astype(np.float64))
    assert (np.abs(y - ref) / np.maximum(ref, 1e-3)).max() < 6 * 2.0 ** -24 and np.all(y >= 0)
    assert float(orc.math_f32("neg_log", np.float32([0.0]))[0]) == float(orc.math_f32("neg_log", np.float32([1.17549435e-38]))[0])
    a, b = rng.permutation(u), u
    y = orc.math_f32("ucomb", a, b)
    ref = (a.astype(np.float64) + b) / (1 + a.astype(np.float64) * b)
    assert (np.abs(y - ref) / np.maximum(ref, 1e-300)).max() < 4 * 2.0 ** -24 and np.all(y <= 1)
    # the identities, in double: u of a product of tanh values, and the logarithm
    x1, x2 = rng.uniform(0.01, 12, 1000), rng.uniform(0.01, 12, 1000)      # (beyond, the double's own 1 - t runs out of digits)
    t = np.tanh(x1 / 2) * np.tanh(x2 / 2)
    uu = (np.exp(-x1) + np.exp(-x2)) / (1 + np.exp(-x1 - x2))
    assert np.allclose((1 - uu) / (1 + uu), t, rtol=1e-12) and np.allclose(-np.log(uu), np.log((1 + t) / (1 - t)), rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("method,schedule,max_iter", [("product_sum", "parallel", 20), ("product_sum", "serial", 6),
                                                      ("minimum_sum", "serial", 6)])
def test_float_edge_form_tracks_double(method, schedule, max_iter):
    """The float per-edge form (what the general HIP kernel computes) against ldpc's arithmetic type: same convergence
    flags and decisions on nearly every shot."""
    H, Lm, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    synd, _, _ = orc.sample_dem(H, Lm, pri, seed=21, shot0=0, B=250)
    g = orc.Graph(H, pri)
    e32, f32 = g.decode_batch(synd, orc.make_params(method, schedule, max_iter, "osd_0", 0, 1.0, orc.FORM_LDPC_F32))
    e64, f64 = g.decode_batch(synd, orc.make_params(method, schedule, max_iter, "osd_0", 0, 1.0, orc.FORM_LDPC_F64))
    assert (f32[:, 0] == f64[:, 0]).mean() > 0.98
    assert (e32 == e64).all(axis=1).mean() > 0.97
    Hd = np.asarray(H.todense(), dtype=np.int64)
    assert np.array_equal(e32.astype(np.int64) @ Hd.T % 2, synd)


def test_grid_arithmetic_is_exact_in_every_form():
    """Channel LLRs on a binary grid (Graph.device_grid: the grid libquits_amd.so picks): min-sum with ms_scaling 1 only adds,
    subtracts, negates and compares, so float or double, ldpc's prefix sums or "total minus own" all return the SAME bits --
    which is why the HIP kernel may be held to the double-precision / ldpc-order form.  Off the grid the float forms differ."""
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    synd, _, _ = orc.sample_dem(H, L, pri, seed=17, shot0=0, B=300)
    forms = (orc.FORM_LDPC_F64, orc.FORM_COMPRESSED_F32, orc.FORM_COMPRESSED_F64, orc.FORM_LDPC_F32)
    g = orc.Graph(H, pri).device_grid(30)
    assert g.grid == orc.grid_bits(pri, 30) and g.grid[1] == g.grid[0] - 4
    out = [g.decode_batch(synd, orc.make_params("minimum_sum", "parallel", 30, "osd_0", 0, 1.0, f), return_grid=True) for f in forms]
    for e, fl, gr in out[1:]:
        assert np.array_equal(e, out[0][0]) and np.array_equal(fl, out[0][1]) and np.array_equal(gr, out[0][2])
    assert (out[0][2][:, 0] == g.grid[0]).all() and not out[0][2][:, 1].any()       # nobody left the fine grid
    raw = orc.Graph(H, pri)
    e64, _ = raw.decode_batch(synd, orc.make_params("minimum_sum", "parallel", 30, "osd_0", 0, 1.0, orc.FORM_LDPC_F64))
    e32, _ = raw.decode_batch(synd, orc.make_params("minimum_sum", "parallel", 30, "osd_0", 0, 1.0, orc.FORM_LDPC_F32))
    assert not np.array_equal(e64, e32)
    # (grid vs exact LLRs: non-converged min-sum is chaotic, the error vectors differ on a quarter of these shots while the
    #  logical error rates agree -- profiles/r02_ler_forms_*.json holds the paired comparison on 2 x 10^6 shots)


def test_grid_bound_sends_a_shot_to_the_coarse_grid():
    """The exactness bound S < 2^(23-k): on a deliberately fine grid it trips, the shot is decoded again on the coarse grid and
    the result is the coarse grid's (what the device's redo pass does); the float forms still agree with the double form."""
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    synd, _, _ = orc.sample_dem(H, L, pri, seed=18, shot0=0, B=120)
    prm = orc.make_params("minimum_sum", "parallel", 40, "osd_0", 0, 1.0, orc.FORM_LDPC_F64)
    g = orc.Graph(H, pri).quantize_llr(17, 10)          # limit 2^6 = 64: most non-trivial shots exceed it
    e, fl, gr = g.decode_batch(synd, prm, return_grid=True)
    tripped = gr[:, 0] == 10
    assert 0 < tripped.sum() < len(synd) and not gr[:, 1].any()
    ec, _ = orc.Graph(H, pri).quantize_llr(10).decode_batch(synd, prm)
    ef, _ = orc.Graph(H, pri).quantize_llr(17).decode_batch(synd, prm)
    assert np.array_equal(e[tripped], ec[tripped]) and np.array_equal(e[~tripped], ef[~tripped])
    e32, fl32, gr32 = g.decode_batch(synd, orc.make_params("minimum_sum", "parallel", 40, "osd_0", 0, 1.0, orc.FORM_COMPRESSED_F32), return_grid=True)
    assert np.array_equal(e32, e) and np.array_equal(gr32, gr)
    assert orc.grid_bits(np.array([0.003]), 50) == (11, 7) and orc.grid_bits(np.array([0.5 - 1e-9]), 1)[0] == 20
    # a large max_iter (ldpc: max_iter = 0 -> n) no longer drags every shot onto a grid of 1/8: the fine grid stops at 2^-10 and
    # the rule's grid becomes the redo grid; the device's 14-bit iteration field caps the rule's max_iter
    assert orc.grid_bits(np.array([0.003]), 9504) == (10, 4) and orc.grid_bits(np.array([0.003]), 200) == (10, 9)
    assert orc.grid_bits(np.array([0.003]), 10 ** 6) == orc.grid_bits(np.array([0.003]), 16383)
    assert orc.device_max_iter(0, 18900) == 16383 and orc.device_max_iter(0, 2592) == 2592 and orc.device_max_iter(7, 100) == 7


def test_lsd0_invariants():
    """The oracle's BP-LSD (LSD-0): every output reproduces its syndrome, the correction lives on the faults the clusters
    took in, a single-fault syndrome whose BP is cut short is repaired by a one-fault cluster, and an inconsistent syndrome is
    flagged instead of looping."""
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    synd, obs, _ = orc.sample_dem(H, L, pri, seed=23, shot0=0, B=300)
    g = orc.Graph(H, pri).device_grid(8)
    prm = orc.make_params("minimum_sum", "parallel", 8, "lsd_0", 0, 1.0, orc.FORM_LDPC_F64)
    err, flags = g.decode_batch(synd, prm)
    Hd = np.asarray(H.todense(), dtype=np.int64)
    assert np.array_equal((err.astype(np.int64) @ Hd.T) % 2, synd)
    assert (flags[:, 0] == 0).sum() > 30                      # LSD really ran
    err_osd, _ = g.decode_batch(synd, orc.make_params("minimum_sum", "parallel", 8, "osd_0", 0, 1.0, orc.FORM_LDPC_F64))
    Ld = np.asarray(L.todense(), dtype=np.int64)
    f_lsd = ((err.astype(np.int64) @ Ld.T) % 2 != obs).any(axis=1).mean()
    f_osd = ((err_osd.astype(np.int64) @ Ld.T) % 2 != obs).any(axis=1).mean()
    assert abs(f_lsd - f_osd) < 0.06, (f_lsd, f_osd)         # same ballpark as OSD-0 on the same posteriors
    # one fault, flat soft information: the cluster seeded at its lowest check takes the lowest-index fault touching it ...
    j = 100
    s1 = np.asarray(H[:, j].todense()).ravel().astype(np.uint8)
    e1, st1 = g.lsd0(s1, np.full(H.shape[1], 3.0))
    assert np.array_equal((e1.astype(np.int64) @ Hd.T) % 2, s1) and st1["added"] >= 1 and not st1["inconsistent"]
    # ... and with the fault marked likely, it is found in one step per seed at most
    llr = np.full(H.shape[1], 3.0); llr[j] = -1.0
    e2, st2 = g.lsd0(s1, llr)
    assert e2[j] == 1 and e2.sum() == 1 and st2["added"] == 1 and st2["pivots"] == 1
    # inconsistent: H of bb144 has rank 1002 < 1008
    H2, _, pri2 = helpers.dem_matrices("bb144_custom_r12_p0.003")
    g2 = orc.Graph(H2, pri2)
    rng = np.random.default_rng(5)
    bad = (rng.random(H2.shape[0]) < 0.4).astype(np.uint8)
    e3, st3 = g2.lsd0(bad, rng.normal(size=H2.shape[1]))
    assert st3["inconsistent"]


def _cost_fixed(err, pri):
    w = np.array([orc.lib().oq_fixed_weight(float(p)) for p in pri], dtype=np.int64)
    return int((err.astype(np.int64) * w).sum())


@pytest.mark.parametrize("method", ["lsd_cs", "lsd_e"])
def test_higher_order_lsd_invariants(method):
    """oq_lsd with lsd_order > 0 (what every BP-LSD call of the reference asks for: tests/test_decoders.py:136): order 0 is
    LSD-0; every output reproduces its syndrome; the growth stage leaves the LSD-0 solution alone, so with integer costs a
    higher order never costs more than LSD-0, and raising the order of 'lsd_e' on the same clusters never costs more; an order
    beyond every cluster's dimension changes nothing further."""
    H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
    synd, obs, _ = orc.sample_dem(H, L, pri, seed=31, shot0=0, B=160)
    g = orc.Graph(H, pri)
    Hd = np.asarray(H.todense(), dtype=np.int64)
    prm = orc.make_params("product_sum", "parallel", 3, "osd_off", 0, 1.0, orc.FORM_LDPC_F64)
    better = grown = 0
    for b in range(160):
        conv, dec, llr, it = g.bp(synd[b], prm)
        if conv:
            continue
        e0, st0 = g.lsd0(synd[b], llr)
        ez, stz = g.lsd(synd[b], llr, method, 0, fixed=True)
        assert np.array_equal(e0, ez) and stz["grown"] == 0 and stz["swept"] == 0
        c0 = _cost_fixed(e0, pri)
        prev = c0
        for order in (1, 2, 5):
            e, st = g.lsd(synd[b], llr, method, order, fixed=True)
            assert np.array_equal((e.astype(np.int64) @ Hd.T) % 2, synd[b]), (b, order)
            c = _cost_fixed(e, pri)
            assert c <= c0, (b, order, c, c0)
            assert st["added"] >= st0["added"] and st["grown"] == st["added"] - st0["added"]
            assert st["replaced"] <= st["swept"]
            better += c < c0
            grown += st["grown"]
            prev = c
        ed, std = g.lsd(synd[b], llr, method, 5, fixed=False)          # ldpc's double costs: same winner unless two costs tie to 1e-5
        assert np.array_equal((ed.astype(np.int64) @ Hd.T) % 2, synd[b])
    assert better > 5 and grown > 20, (better, grown)


def test_higher_order_lsd_small_case_by_brute_force():
    """One cluster, hand-checkable: repetition-code chain H (4 checks x 6 faults incl. two parallel faults on check 1-2).
    Syndrome on checks 1, 2.  LSD-0 takes the lowest-LLR fault; with order 1 the cheaper parallel fault must win when costs say so."""
    from scipy.sparse import csc_matrix
    H = csc_matrix(np.array([[1, 1, 0, 0, 0, 0],
                             [0, 1, 1, 1, 0, 0],
                             [0, 0, 1, 1, 1, 0],
                             [0, 0, 0, 0, 1, 1]], dtype=np.uint8))
    pri = np.array([0.01, 0.01, 0.001, 0.2, 0.01, 0.01])          # fault 3 is cheap (log 1/p small), fault 2 expensive
    g = orc.Graph(H, pri)
    s = np.array([0, 1, 1, 0], np.uint8)
    llr = np.array([3.0, 3.0, 0.5, 1.0, 3.0, 3.0])                # posteriors prefer fault 2
    e0, st0 = g.lsd0(s, llr)
    assert e0.tolist() == [0, 0, 1, 0, 0, 0] and st0["added"] == 1
    e1, st1 = g.lsd(s, llr, "lsd_cs", 1)
    # growth stage: dimension 0 < 1 -> one more fault joins (lowest LLR touching checks 1, 2: fault 3), it is dependent on fault 2
    # (same column) -> non-pivot; sweep: flip fault 3 => fault 2 off; cost log(1/0.2) < log(1/0.001)
    assert st1["grown"] == 1 and st1["swept"] == 1 and st1["replaced"] == 1
    assert e1.tolist() == [0, 0, 0, 1, 0, 0]
    # with the costs the other way round the LSD-0 solution stays
    g2 = orc.Graph(H, np.array([0.01, 0.01, 0.2, 0.001, 0.01, 0.01]))
    e2, st2 = g2.lsd(s, llr, "lsd_cs", 1)
    assert e2.tolist() == [0, 0, 1, 0, 0, 0] and st2["replaced"] == 0


def test_native_build_matches_portable(tmp_path):
    """bench.py's cpu_baseline leg times a `-O3 -march=native` build of the port made on the host it runs on (oracle/_native/, never
    shipped).  With -ffp-contract=off the two builds must agree bit for bit: decode the same shots with both (the native one in a
    child process, since a process loads one build)."""
    import subprocess
    import sys
    name = "bb72_custom_r6_p0.003"
    H, L, pri = helpers.dem_matrices(name)
    rng = np.random.default_rng(5)
    e = (rng.random((48, H.shape[1])) < pri).astype(np.uint8)
    synd = np.ascontiguousarray((csr_matrix(H) @ e.T % 2).T.astype(np.uint8))
    np.save(tmp_path / "synd.npy", synd)
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import helpers, oracle as orc
native = sys.argv[2] == "1"
orc.use_native(native)
H, L, pri = helpers.dem_matrices(%r)
g = orc.Graph(H, pri)
out = []
for bp, sch, osd, order, form in (("minimum_sum", "parallel", "osd_0", 0, orc.FORM_LDPC_F64), ("product_sum", "serial", "osd_cs", 1, orc.FORM_LDPC_F64),
                                  ("product_sum", "parallel", "osd_0", 0, orc.FORM_LDPC_F32), ("minimum_sum", "serial", "lsd_cs", 1, orc.FORM_LDPC_F32)):
    dec, st = g.decode_batch(np.load(sys.argv[1]), orc.make_params(bp, sch, 12, osd, order, 1.0, form))
    out.append(dec)
assert orc.build_info()["native"] == native, orc.build_info()
np.save(sys.argv[3], np.stack(out))
''' % (os.path.dirname(os.path.abspath(orc.__file__)), os.path.dirname(os.path.abspath(helpers.__file__)), name)
    outs = []
    for native in ("0", "1"):
        dst = str(tmp_path / ("out%s.npy" % native))
        subprocess.check_call([sys.executable, "-c", code, str(tmp_path / "synd.npy"), native, dst])
        outs.append(np.load(dst))
    assert np.array_equal(outs[0], outs[1])
