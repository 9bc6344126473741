#!/usr/bin/env python3
"""Generate tests/golden/ by importing the reference (read-only, stubbed stim/ldpc) in THIS container.

Run:  python -B tools/gen_fixtures.py [all|codes|dem|windows|loop|gf2|qlp]
The reference cannot travel to the GPU box, so everything the tests need from it is captured here as
small data fixtures (inputs + expected outputs).  Nothing under tests/ or quits_amd/ imports the reference.

G1  code matrices hz/hx/lz/lx (bit-packed)                 <- quits.qldpc_code.{HgpCode,BbCode,QlpCode}
G2  circuit text of BASELINE.json configs (gzip)           <- code.build_circuit(...)
G3  reference spacetime() outputs on this package's DEM     <- quits.decoder.base.spacetime
G4  reference detector_error_model_to_matrix on a hand-made DEM with duplicate symptoms
G5  reference sliding_window_{circuit,phenom}_mem outputs with a deterministic plug-in decoder
G6  reference gf2_solve / gf2_rank / gf2_rref known answers <- quits.gf2_util
G7  window-count table (num_cor_rounds, W_last) for an (R, W, F) grid
"""
import gzip
import io
import json
import os
import sys
import warnings

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from _refimport import import_reference  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def _pack(a):
    a = np.asarray(a, dtype=np.uint8)
    return {"shape": np.asarray(a.shape, np.int64), "bits": np.packbits(a, axis=None)}


def _save_code(name, code):
    out = {}
    for k in ("hz", "hx", "lz", "lx"):
        p = _pack(getattr(code, k))
        out[k + "_shape"] = p["shape"]
        out[k + "_bits"] = p["bits"]
    np.savez_compressed(os.path.join(GOLD, "codes", name + ".npz"), **out)


def _save_text(name, text):
    with gzip.GzipFile(os.path.join(GOLD, "circuits", name + ".stim.gz"), "wb", mtime=0) as f:
        f.write(str(text).encode())


def gen_codes(with_qlp=False):
    import_reference()
    from quits.noise import ErrorModel
    from quits.qldpc_code import BbCode, HgpCode, QlpCode

    os.makedirs(os.path.join(GOLD, "codes"), exist_ok=True)
    os.makedirs(os.path.join(GOLD, "circuits"), exist_ok=True)
    meta = {}
    # config 1: HGP from the shipped n=12 (3,4)-LDPC matrix, cardinal, seed 1, R=3, p=0.01
    h = np.loadtxt(os.path.join(REF, "parity_check_matrices", "n=12_dv=3_dc=4_dist=6.txt"), dtype=int)
    np.save(os.path.join(GOLD, "codes", "ldpc_n12_dv3_dc4.npy"), h.astype(np.uint8))
    hgp = HgpCode(h, h)
    _save_code("hgp225", hgp)
    c = hgp.build_circuit(strategy="cardinal", error_model=ErrorModel(0.01, 0.01, 0.01, 0.01), num_rounds=3,
                          basis="Z", seed=1)
    _save_text("hgp225_cardinal_r3_p0.01", c)
    meta["hgp225_cardinal_r3_p0.01"] = dict(code="hgp225", rounds=3, p=0.01, strategy="cardinal", seed=1)
    # configs 2-4: BB codes, custom circuit (BB does not support 'cardinal': bb.py:20)
    for name, (l, m), R, ps in (("bb72", (6, 6), 6, (0.003,)),
                                ("bb144", (12, 6), 12, (0.001, 0.002, 0.003, 0.004, 0.005, 0.006)),
                                ("bb90", (15, 3), 15, (0.001,))):
        if name == "bb90":
            code = BbCode(15, 3, [9], [1, 2], [2, 7], [0])
        else:
            code = BbCode(l, m, [3], [1, 2], [1, 2], [3])
        _save_code(name, code)
        for p in ps:
            c = code.build_circuit(strategy="custom", error_model=ErrorModel(p, p, p, p), num_rounds=R, basis="Z")
            key = "%s_custom_r%d_p%g" % (name, R, p)
            _save_text(key, c)
            meta[key] = dict(code=name, rounds=R, p=p, strategy="custom")
    # a small BB72 X-basis + 2-round circuit for parser edge cases
    code = BbCode(6, 6, [3], [1, 2], [1, 2], [3])
    c = code.build_circuit(strategy="custom", error_model=ErrorModel(0.002, 0.001, 0.003, 0.004), num_rounds=2, basis="X")
    _save_text("bb72_custom_r2_xbasis_mixed", c)
    meta["bb72_custom_r2_xbasis_mixed"] = dict(code="bb72", rounds=2, p=[0.002, 0.001, 0.003, 0.004], basis="X")
    # every CircuitBuildOptions switch away from its default (circuit_build_options.py:13-15): X and Z detectors together,
    # noiseless zeroth round, noisy final measurement
    from quits.qldpc_code.circuit_construction.circuit_build_options import CircuitBuildOptions
    c = code.build_circuit(strategy="custom", error_model=ErrorModel(0.003, 0.003, 0.003, 0.003), num_rounds=2, basis="Z",
                           circuit_build_options=CircuitBuildOptions(get_all_detectors=True, noisy_zeroth_round=False,
                                                                     noisy_final_meas=True))
    _save_text("bb72_custom_r2_alldet_p0.003", c)
    meta["bb72_custom_r2_alldet_p0.003"] = dict(code="bb72", rounds=2, p=0.003, basis="Z", get_all_detectors=True,
                                                noisy_zeroth_round=False, noisy_final_meas=True)
    c = code.build_circuit(strategy="custom", error_model=ErrorModel(0.003, 0.003, 0.003, 0.003), num_rounds=0, basis="Z")
    _save_text("bb72_custom_r0_p0.003", c)
    meta["bb72_custom_r0_p0.003"] = dict(code="bb72", rounds=0, p=0.003)
    if with_qlp:
        b = np.array([[0, 0, 0, 0, 0], [0, 2, 14, 24, 25], [0, 16, 11, 14, 13]])
        q = QlpCode(b, b, 30)
        _save_code("qlp1020", q)
        c = q.build_circuit(strategy="cardinal", error_model=ErrorModel(0.003, 0.003, 0.003, 0.003), num_rounds=20,
                            basis="Z", seed=1)
        _save_text("qlp1020_cardinal_r20_p0.003", c)
        meta["qlp1020_cardinal_r20_p0.003"] = dict(code="qlp1020", rounds=20, p=0.003, strategy="cardinal", seed=1)
    mpath = os.path.join(GOLD, "circuits", "index.json")
    old = {}
    if os.path.exists(mpath):
        old = json.load(open(mpath))
    old.update(meta)
    json.dump(old, open(mpath, "w"), indent=1, sort_keys=True)
    print("codes/circuits written:", sorted(meta))


# ----------------------------------------------------------------------------------------------------------
def gen_anchor_circuits():
    """Circuits of the reference's EXECUTED notebook cells, whose printed outputs are the only results of the real
    ldpc + Stim pipeline available to this repo (doc/06A_end_to_end_demo_hgp.ipynb cell 5, doc/06B_end_to_end_demo_bb.ipynb
    cell 5, doc/04_decoding_sliding_window.ipynb cells 5-9, doc/05_decoder_variants.ipynb cells 8-9,
    doc/00_getting_started.ipynb cell 8).  One fixture per circuit at p = 1e-3; the other rates of a sweep differ only by the
    printed '%.10f' literal (tests/helpers.py: circuit_text_at_p)."""
    import_reference()
    from quits.noise import ErrorModel
    from quits.qldpc_code import BbCode, HgpCode
    from quits.qldpc_code.circuit_construction.circuit_build_options import CircuitBuildOptions

    meta = {}
    h = np.loadtxt(os.path.join(REF, "parity_check_matrices", "n=12_dv=3_dc=4_dist=6.txt"), dtype=int)
    hgp = HgpCode(h, h)
    p = 1e-3
    c = hgp.build_circuit(error_model=ErrorModel(p, p, p, p), num_rounds=15, basis="Z",
                          circuit_build_options=CircuitBuildOptions(), seed=1)      # 06A cell 5 / 04 cell 5: default strategy
    assert hgp.depth == 8, hgp.depth                                               # "# layer of entangling gates:  8" (04 cell 5 output)
    _save_text("hgp225_cardinal_r15_p0.001", c)
    meta["hgp225_cardinal_r15_p0.001"] = dict(code="hgp225", rounds=15, p=p, strategy="default(cardinal)", seed=1, depth=8)
    bb = BbCode(l=15, m=3, A_x_pows=[9], A_y_pows=[1, 2], B_x_pows=[2, 7], B_y_pows=[0])
    c = bb.build_circuit(error_model=ErrorModel(p, p, p, p), num_rounds=15, basis="Z",
                         circuit_build_options=CircuitBuildOptions())              # 06B cell 5: default strategy
    old = _load_text("bb90_custom_r15_p0.001")
    assert str(c) == old, "06B's default-strategy circuit differs from the committed custom-strategy fixture"
    # 00_getting_started cell 4/8: 3 x 3 cyclic repetition matrix -> HGP [[18,2,3]], zxcoloration, R = 3
    H = np.zeros((3, 3), dtype=int)
    for i in range(3):
        H[i, i] = 1
        H[i, (i + 1) % 3] = 1
    small = HgpCode(H, H)
    _save_code("hgp_rep3", small)
    c = small.build_circuit(strategy="zxcoloration", error_model=ErrorModel(idle_error=1e-3, sqgate_error=1e-3,
                                                                            tqgate_error=1e-3, spam_error=1e-3),
                            num_rounds=3, basis="Z")
    assert small.depth == 8, small.depth                                           # "HGP zxcoloration depth: 8" (00 cell 4 output)
    _save_text("hgprep3_zxcoloration_r3_p0.001", c)
    meta["hgprep3_zxcoloration_r3_p0.001"] = dict(code="hgp_rep3", rounds=3, p=1e-3, strategy="zxcoloration", depth=8)
    # the reference's own decoder tests (tests/test_decoders.py:9-31,88-159): BPC code, cardinal seed 1, R = 10, p = 5e-4
    from quits.qldpc_code import BpcCode
    bpc = BpcCode([0, 1, 5], [0, 8, 13], 15, 3)
    c = bpc.build_circuit(strategy="cardinal", error_model=ErrorModel(5e-4, 5e-4, 5e-4, 5e-4), num_rounds=10, basis="Z", seed=1)
    _save_code("bpc_15_3", bpc)
    _save_text("bpc_cardinal_r10_p0.0005", c)
    meta["bpc_cardinal_r10_p0.0005"] = dict(code="bpc_15_3", rounds=10, p=5e-4, strategy="cardinal", seed=1, depth=int(bpc.depth))
    mpath = os.path.join(GOLD, "circuits", "index.json")
    idx = json.load(open(mpath))
    idx.update(meta)
    json.dump(idx, open(mpath, "w"), indent=1, sort_keys=True)
    print("anchor circuits written:", sorted(meta), "hgp_rep3 hz", small.hz.shape, "lz", small.lz.shape)


def _load_text(name):
    with gzip.open(os.path.join(GOLD, "circuits", name + ".stim.gz"), "rb") as f:
        return f.read().decode()


def _load_code(name):
    z = np.load(os.path.join(GOLD, "codes", name + ".npz"))
    out = {}
    for k in ("hz", "hx", "lz", "lx"):
        shp = tuple(z[k + "_shape"])
        out[k] = np.unpackbits(z[k + "_bits"])[: shp[0] * shp[1]].reshape(shp)
    return out


def _csc_dump(prefix, mat, out):
    from scipy.sparse import csc_matrix
    mat = csc_matrix(mat)
    mat.sort_indices()
    out[prefix + "_shape"] = np.asarray(mat.shape, np.int64)
    out[prefix + "_indptr"] = mat.indptr.astype(np.int32)
    out[prefix + "_indices"] = mat.indices.astype(np.int32)


def gen_windows(which="std"):
    """G3 + G7: reference spacetime() on the DEM produced by quits_amd.dem (duck-typed).  `qlp`: BASELINE configs[4]
    (QLP [[1020,136]], R = 20, W = 3, F = 1: 20 windows of 1350 x 18900)."""
    import_reference()
    from quits.decoder.base import detector_error_model_to_matrix, spacetime
    from quits_amd.dem import Circuit

    os.makedirs(os.path.join(GOLD, "windows"), exist_ok=True)
    sets = (("bb72_custom_r6_p0.003", "bb72", 6, ((3, 1), (5, 3), (8, 1), (4, 2))),
            ("bb144_custom_r12_p0.003", "bb144", 12, ((3, 1), (5, 3))),
            ("hgp225_cardinal_r3_p0.01", "hgp225", 3, ((3, 1), (2, 1))))
    if which == "qlp":
        sets = (("qlp1020_cardinal_r20_p0.003", "qlp1020", 20, ((3, 1),)),)
    for cname, code, R, grid in sets:
        circ = Circuit(_load_text(cname))
        hz = _load_code(code)["hz"]
        H, L, pri = detector_error_model_to_matrix(circ.detector_error_model())
        out = {}
        _csc_dump("H", H, out)
        _csc_dump("L", L, out)
        out["priors"] = pri
        for (W, F) in grid:
            if 2 + R - W >= 0:
                ncr = (2 + R - W) // F + (1 if (2 + R - W) % F else 0)
            else:
                ncr = 0
            a, b, c, d = spacetime(circ, hz, W, F, ncr)
            tag = "W%dF%d" % (W, F)
            out[tag + "_nwin"] = np.asarray([len(a)], np.int64)
            for k in range(len(a)):
                _csc_dump("%s_H%d" % (tag, k), a[k], out)
                _csc_dump("%s_L%d" % (tag, k), b[k], out)
                out["%s_p%d" % (tag, k)] = np.asarray(c[k], np.float64)
                if k < len(d):
                    _csc_dump("%s_U%d" % (tag, k), d[k], out)
        np.savez_compressed(os.path.join(GOLD, "windows", cname + ".npz"), **out)
        print("windows:", cname, H.shape, H.nnz)
    if which == "qlp":
        return
    # G7: window-count table straight from the reference's arithmetic (sliding_window.py:134-141) by running it
    rows = []
    from quits.decoder.sliding_window import sliding_window_phenom_mem

    class _Zero:
        def __init__(self, pcm, **kw):
            self.n = pcm.shape[1]
            type(self).shapes.append(pcm.shape)

        def decode(self, s):
            return np.zeros(self.n, dtype=int)

    hz = np.array([[1, 1, 0], [0, 1, 1]])
    lz = np.array([[1, 1, 1]])
    for R in range(0, 9):
        for W in range(1, 8):
            for F in range(1, W + 1):
                _Zero.shapes = []
                with warnings.catch_warnings(record=True) as wlog:
                    warnings.simplefilter("always")
                    sliding_window_phenom_mem(np.zeros((1, 2 * (R + 2)), dtype=int), hz, lz, W, F, _Zero, _Zero,
                                              {}, {}, "decode", "decode")
                rows.append((R, W, F, _Zero.shapes[0][0] // 2, _Zero.shapes[1][0] // 2, int(len(wlog) > 0)))
    np.save(os.path.join(GOLD, "windows", "window_table.npy"), np.asarray(rows, np.int64))
    print("window table rows:", len(rows))


def gen_dem_merge():
    """G4: reference detector_error_model_to_matrix on a hand-made DEM with duplicate detector sets."""
    import_reference()
    from quits.decoder.base import detector_error_model_to_matrix
    from quits_amd.dem import DetectorErrorModel

    errors = [(0.01, (0, 1), ()), (0.02, (1, 2), (0,)), (0.03, (1, 0), (1,)), (0.04, (3,), ()),
              (0.05, (1, 2), ()), (0.06, (2, 3, 4), (0, 1)), (0.07, (3,), (1,)), (0.08, (4,), ()),
              (0.09, (0, 1), (0,)), (0.1, (0, 4), ())]
    dem = DetectorErrorModel(errors, 5, 2)
    buf = io.StringIO()
    so = sys.stdout
    sys.stdout = buf
    try:
        H, L, pri = detector_error_model_to_matrix(dem)
    finally:
        sys.stdout = so
    out = {"err_p": np.asarray([e[0] for e in errors]),
           "err_dets": np.asarray([",".join(map(str, e[1])) for e in errors]),
           "err_obs": np.asarray([",".join(map(str, e[2])) for e in errors]),
           "num_detectors": np.asarray([5]), "num_observables": np.asarray([2]), "priors": pri}
    _csc_dump("H", H, out)
    _csc_dump("L", L, out)
    np.savez_compressed(os.path.join(GOLD, "dem_merge.npz"), **out)
    print("dem merge:", H.shape, pri)


def gen_loop():
    """G5: the reference's own per-shot loops (sliding_window.py:14-188) driven by this repo's deterministic CPU
    oracle decoder as the plug-in, on DEM-sampled syndromes -> expected logical predictions."""
    import_reference()
    from quits.decoder.sliding_window import sliding_window_circuit_mem, sliding_window_phenom_mem
    from quits_amd.dem import Circuit
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    from quits.decoder.base import detector_error_model_to_matrix

    os.makedirs(os.path.join(GOLD, "loop"), exist_ok=True)
    for cname, code, R, N, cases in (
            ("bb72_custom_r6_p0.003", "bb72", 6, 192, ((3, 1, 20), (5, 3, 12), (8, 1, 30), (9, 2, 30))),
            ("hgp225_cardinal_r3_p0.01", "hgp225", 3, 48, ((3, 1, 15), (2, 1, 15))),
            ("bb72_custom_r2_xbasis_mixed", "bb72", 2, 96, ((2, 1, 15), (4, 1, 15), (3, 2, 15)))):      # X-basis memory: hx, lx
        circ = Circuit(_load_text(cname))
        cd = _load_code(code)
        hz, lz = (cd["hx"], cd["lx"]) if "xbasis" in cname else (cd["hz"], cd["lz"])
        H, L, pri = detector_error_model_to_matrix(circ.detector_error_model())
        synd, obs, _ = orc.sample_dem(H, L, pri, seed=20260929, shot0=0, B=N)
        out = {"syndromes": np.packbits(synd, axis=1), "observables": np.packbits(obs, axis=1),
               "shape": np.asarray(synd.shape, np.int64)}
        for (W, F, mi) in cases:
            # "grid": ldpc's double-precision arithmetic on the LLR grid the HIP library uses (exact there; see
            # qd_decoder_info in include/quits_amd.h) -- what the device path must reproduce bit for bit
            for grid, ftag in ((None, "f64"), ("device", "grid")):
                d1 = dict(bp_method="minimum_sum", max_iter=mi, schedule="parallel", osd_method="osd_0",
                          osd_order=0, form=orc.FORM_LDPC_F64, llr_grid=grid)
                d2 = dict(d1)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    pred = sliding_window_circuit_mem(synd.astype(int), circ, hz, lz, W, F,
                                                      orc.OracleBpOsdDecoder, orc.OracleBpOsdDecoder, d1, d2,
                                                      "channel_probs", "channel_probs", "decode", "decode")
                out["circ_W%dF%d_it%d_%s" % (W, F, mi, ftag)] = pred.astype(np.uint8)
        # phenomenological loop on the same detector record (the reference tests do exactly this,
        # tests/test_sliding_window.py:106-166), scalar prior
        for (W, F, mi) in cases[:2]:
            d1 = dict(bp_method="minimum_sum", max_iter=mi, schedule="parallel", osd_method="osd_0", osd_order=0,
                      error_rate=0.03, form=orc.FORM_LDPC_F64, llr_grid="device")
            d2 = dict(d1)
            pred = sliding_window_phenom_mem(synd.astype(int), hz, lz, W, F, orc.OracleBpOsdDecoder,
                                             orc.OracleBpOsdDecoder, d1, d2, "decode", "decode")
            out["phen_W%dF%d_it%d_grid" % (W, F, mi)] = pred.astype(np.uint8)
        np.savez_compressed(os.path.join(GOLD, "loop", cname + ".npz"), **out)
        print("loop:", cname, {k: v.shape for k, v in out.items() if k.startswith(("circ", "phen"))})


def gen_codecap():
    """G8: the reference's code-capacity driver (simulation.py:31-61) run with the oracle decoder as plug-in ->
    logical error rates for fixed seeds."""
    import json
    import types
    import_reference()
    from quits.simulation import get_codecap_pL
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    out = []
    for code, p, trials, seed, basis, opts in (
            ("hgp225", 0.02, 300, 1, "Z", dict(bp_method="minimum_sum", schedule="parallel", max_iter=20, osd_method="osd_0", osd_order=0)),
            ("hgp225", 0.03, 200, 5, "X", dict(bp_method="product_sum", schedule="serial", max_iter=4, osd_method="osd_cs", osd_order=2)),
            ("bb72", 0.04, 300, 2, "Z", dict(bp_method="minimum_sum", schedule="parallel", max_iter=15, osd_method="osd_0", osd_order=0)),
            ("bb72", 0.05, 200, 9, "Z", dict(bp_method="product_sum", schedule="parallel", max_iter=10, osd_method="osd_0", osd_order=0))):
        cd = _load_code(code)
        cobj = types.SimpleNamespace(hz=cd["hz"], hx=cd["hx"], lz=cd["lz"], lx=cd["lx"])
        for form, ftag in ((orc.FORM_LDPC_F64, "f64"), (orc.FORM_LDPC_F32, "f32")):
            d = dict(opts, error_rate=p, form=form)
            if opts["bp_method"] == "minimum_sum" and opts["schedule"] == "parallel" and ftag == "f32":
                ftag = "grid"
                d = dict(opts, error_rate=p, form=orc.FORM_LDPC_F64, llr_grid="device")
            pl = get_codecap_pL(cobj, p, trials, orc.OracleBpOsdDecoder, d, basis=basis, seed=seed)
            out.append(dict(code=code, p=p, trials=trials, seed=seed, basis=basis, opts=opts, form=ftag, pL=pl))
            print("codecap:", code, p, basis, opts["bp_method"], opts["schedule"], ftag, pl)
    json.dump(out, open(os.path.join(GOLD, "codecap.json"), "w"), indent=1)


def gen_gf2():
    """G6: known answers from the reference's dense GF(2) algebra (gf2_util.py:20,51,146)."""
    import_reference()
    from quits.gf2_util import gf2_rank, gf2_solve

    rng = np.random.default_rng(7)
    out = {}
    idx = 0
    for (m, n, dens) in ((6, 9, 0.4), (12, 30, 0.2), (40, 90, 0.08), (64, 64, 0.1), (65, 130, 0.05), (30, 20, 0.2),
                         (100, 300, 0.03)):
        for rep in range(3):
            A = (rng.random((m, n)) < dens).astype(np.uint8)
            if rep == 2 and m > 4:
                A[m - 1] = A[0] ^ A[1]       # force a row dependency
            x = (rng.random(n) < 0.2).astype(np.uint8)
            b = (A @ x % 2).astype(np.uint8)
            r = int(gf2_rank(A))
            sol = gf2_solve(A, b)
            assert sol is not None and ((A @ np.asarray(sol).reshape(-1) % 2) == b).all()
            b_bad = b.copy()
            consistent_bad = None
            if r < m:
                # an inconsistent right-hand side, if one exists cheaply
                for tr in range(m):
                    bb = b.copy(); bb[tr] ^= 1
                    if gf2_solve(A, bb) is None:
                        b_bad = bb; consistent_bad = 0
                        break
            out["A%d" % idx] = np.packbits(A, axis=1)
            out["shape%d" % idx] = np.asarray([m, n])
            out["b%d" % idx] = b
            out["rank%d" % idx] = np.asarray([r])
            out["bbad%d" % idx] = b_bad
            out["bbad_ok%d" % idx] = np.asarray([-1 if consistent_bad is None else 0])
            idx += 1
    out["count"] = np.asarray([idx])
    np.savez_compressed(os.path.join(GOLD, "gf2.npz"), **out)
    print("gf2 systems:", idx)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    os.makedirs(GOLD, exist_ok=True)
    if what in ("all", "codes"):
        gen_codes(with_qlp=False)
    if what == "qlp":
        gen_codes(with_qlp=True)
    if what in ("all", "anchor"):
        gen_anchor_circuits()
    if what in ("all", "dem"):
        gen_dem_merge()
    if what in ("all", "windows"):
        gen_windows()
    if what == "qlpwin":
        gen_windows("qlp")
    if what in ("all", "gf2"):
        gen_gf2()
    if what in ("all", "loop"):
        gen_loop()
    if what in ("all", "codecap"):
        gen_codecap()
    left = [p for p, _, _ in os.walk(REF) if p.endswith("__pycache__")]
    assert not left, "bytecode was written under /root/reference: %r" % left
