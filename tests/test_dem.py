"""Circuit text -> detector error model (quits_amd/stim_text.py, quits_amd/dem.py): CPU tests."""
import numpy as np
import pytest

import helpers
from quits_amd.dem import Circuit, DetectorErrorModel, circuit_to_dem
from quits_amd.decoder.base import detector_error_model_to_matrix
from quits_amd.stim_text import CircuitSyntaxError, flatten

EXPECTED = {   # detectors, DEM columns, nnz
    "hgp225_cardinal_r3_p0.01": (540, 5409, 22356, 9),
    "bb72_custom_r6_p0.003": (288, 2592, 9036, 12),
    "bb144_custom_r12_p0.003": (1008, 9504, 33192, 12),
}


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_dem_shapes_and_golden_matrices(name):
    """H, L, priors equal the arrays the REFERENCE's detector_error_model_to_matrix produced from this DEM
    (tests/golden/windows/*.npz, tools/gen_fixtures.py G3) -- pins the restated function bit for bit."""
    dem = Circuit(helpers.circuit_text(name)).detector_error_model()
    det, cols, nnz, nobs = EXPECTED[name]
    assert (dem.num_detectors, dem.num_errors, dem.num_observables) == (det, cols, nobs)
    H, L, pri = detector_error_model_to_matrix(dem)
    Hg, Lg, pg = helpers.dem_matrices(name)
    assert H.nnz == nnz and helpers.same_sparse(H, Hg) and helpers.same_sparse(L, Lg)
    assert np.array_equal(pri, pg)
    assert H.dtype == np.uint8 and pri.dtype == np.float64


def test_merge_rule_matches_reference():
    """G4: duplicate detector sets fold probabilities, first-seen column order, first-seen observables kept."""
    z = np.load(helpers.GOLD + "/dem_merge.npz")
    errors = []
    for p, d, o in zip(z["err_p"], z["err_dets"], z["err_obs"]):
        errors.append((float(p), tuple(int(x) for x in str(d).split(",") if x), tuple(int(x) for x in str(o).split(",") if x)))
    dem = DetectorErrorModel(errors, int(z["num_detectors"][0]), int(z["num_observables"][0]))
    H, L, pri = detector_error_model_to_matrix(dem)
    assert helpers.same_sparse(H, helpers.csc_from(z, "H")) and helpers.same_sparse(L, helpers.csc_from(z, "L"))
    assert np.allclose(pri, z["priors"], rtol=0, atol=0)


def test_priors_and_ordering_bb144():
    dem = Circuit(helpers.circuit_text("bb144_custom_r12_p0.003")).detector_error_model()
    ps = np.array([e[0] for e in dem.errors])
    assert abs(ps.sum() - 51.612) < 1e-2 and ps.min() > 1.5e-3 and ps.max() < 1.7e-2
    keys = [tuple(d) + tuple(dem.num_detectors + o for o in ob) for (_, d, ob) in dem.errors]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    # every fault touches at most two consecutive rounds of 72 detectors (what spacetime() relies on)
    for (_, d, _) in dem.errors:
        assert max(d) // 72 - min(d) // 72 <= 1


def _forward_symptom(ops, num_det_of_meas, start, pauli):
    """Independent check: push one Pauli fault FORWARD through the circuit and collect the flipped measurements."""
    x, z = dict(), dict()
    for q, p in pauli.items():
        if p in "XY":
            x[q] = 1
        if p in "ZY":
            z[q] = 1
    flips = 0
    m = sum(len(op.targets) for op in ops[:start] if op.name in ("M", "MX", "MR"))
    for op in ops[start:]:
        t = op.targets
        if op.name == "CX":
            for i in range(0, len(t), 2):
                c, tg = t[i], t[i + 1]
                if x.get(c):
                    x[tg] = x.get(tg, 0) ^ 1
                if z.get(tg):
                    z[c] = z.get(c, 0) ^ 1
        elif op.name == "H":
            for q in t:
                x[q], z[q] = z.get(q, 0), x.get(q, 0)
        elif op.name in ("M", "MR", "MX"):
            for q in t:
                hit = z.get(q, 0) if op.name == "MX" else x.get(q, 0)
                if hit:
                    flips ^= num_det_of_meas[m]
                m += 1
                if op.name == "MR":
                    x[q] = 0
                    z[q] = 0
        elif op.name in ("R", "RX"):
            for q in t:
                x[q] = 0
                z[q] = 0
    return flips


@pytest.mark.parametrize("name", ["bb72_custom_r0_p0.003", "bb72_custom_r2_xbasis_mixed", "bb72_custom_r2_alldet_p0.003"])
def test_backward_extractor_against_forward_propagation(name):
    """Every fault component of every noise instruction, propagated forward by an independent Pauli-frame walk,
    must land on a DEM column with that symptom, and the merged probabilities must agree."""
    text = helpers.circuit_text(name)
    ops, nmeas, ndet, nobs = flatten(text)
    sens = [0] * nmeas
    for op in ops:
        if op.name == "DETECTOR":
            for k in op.targets:
                sens[k] ^= 1 << int(op.arg)
        elif op.name == "OBSERVABLE_INCLUDE":
            for k in op.targets:
                sens[k] ^= 1 << (ndet + int(op.arg))
    probs = {}

    def add(sym, q):
        if sym:
            p = probs.get(sym, 0.0)
            probs[sym] = p * (1 - q) + q * (1 - p)

    for i, op in enumerate(ops):
        if op.name == "X_ERROR":
            for q in op.targets:
                add(_forward_symptom(ops, sens, i + 1, {q: "X"}), op.arg)
        elif op.name == "Z_ERROR":
            for q in op.targets:
                add(_forward_symptom(ops, sens, i + 1, {q: "Z"}), op.arg)
        elif op.name == "DEPOLARIZE1":
            q1 = 0.5 - 0.5 * np.sqrt(1 - 4 * op.arg / 3)
            for q in op.targets:
                for P in "XYZ":
                    add(_forward_symptom(ops, sens, i + 1, {q: P}), q1)
        elif op.name == "DEPOLARIZE2":
            q2 = 0.5 - 0.5 * (1 - 16 * op.arg / 15) ** 0.125
            t = op.targets
            for j in range(0, len(t), 2):
                for Pa in "IXYZ":
                    for Pb in "IXYZ":
                        if Pa + Pb != "II":
                            pl = {}
                            if Pa != "I":
                                pl[t[j]] = Pa
                            if Pb != "I":
                                pl[t[j + 1]] = Pb
                            add(_forward_symptom(ops, sens, i + 1, pl), q2)
    dem = circuit_to_dem(text)
    got = {}
    for (p, d, o) in dem.errors:
        got[sum(1 << x for x in d) | sum(1 << (ndet + x) for x in o)] = p
    assert set(got) == set(probs)
    for k in probs:
        assert abs(got[k] - probs[k]) < 1e-15


def test_parser_edge_cases():
    ops, nm, nd, no = flatten("R 0 1\nREPEAT 3 {\n  H 0\n  CX 0 1\n  M 1\n  DETECTOR rec[-1]\n}\nM 0\nOBSERVABLE_INCLUDE(2) rec[-1] rec[-2]\n")
    assert nm == 4 and nd == 3 and no == 3
    assert [o.name for o in ops].count("CX") == 3
    assert ops[-1].targets == (3, 2)
    with pytest.raises(CircuitSyntaxError):
        flatten("REPEAT 2 {\nH 0\n")
    with pytest.raises(CircuitSyntaxError):
        flatten("M 0\nDETECTOR rec[-2]\n")
    with pytest.raises(CircuitSyntaxError):
        flatten("FROBNICATE 0\n")
    with pytest.raises(NotImplementedError):
        flatten("PAULI_CHANNEL_1(0.1, 0.2, 0.3) 0\n")
    with pytest.raises(CircuitSyntaxError):
        flatten("CX 0 1 2\n")
    assert circuit_to_dem("R 0\nM 0\nDETECTOR rec[-1]\n").num_errors == 0     # noiseless: empty DEM
    d = circuit_to_dem("R 0\nX_ERROR(0.125) 0\nM 0\nDETECTOR rec[-1]\n")
    assert d.errors == [(0.125, (0,), ())]
    d = circuit_to_dem("RX 0\nZ_ERROR(0.25) 0\nX_ERROR(0.5) 0\nMX 0\nDETECTOR rec[-1]\nOBSERVABLE_INCLUDE(0) rec[-1]\n")
    assert d.errors == [(0.25, (0,), (0,))]


def test_circuit_wrapper_api():
    c = Circuit(helpers.circuit_text("bb72_custom_r6_p0.003"))
    assert c.num_detectors == 288 and c.num_observables == 12
    assert c.detector_error_model() is c.detector_error_model(decompose_errors=False)
    with pytest.raises(NotImplementedError):
        c.detector_error_model(decompose_errors=True)
    inst = c.detector_error_model().flattened()[0]
    assert inst.type == "error" and len(inst.args_copy()) == 1
    t = inst.targets_copy()
    assert t[0].is_relative_detector_id() and not t[0].is_logical_observable_id()


def test_rate_substitution_reproduces_the_reference_circuit():
    base = "bb144_custom_r12_p0.003"
    for p in (0.001, 0.002, 0.004, 0.005, 0.006):
        assert helpers.circuit_text_at_p(base, 0.003, p) == helpers.circuit_text("bb144_custom_r12_p%g" % p)


def test_as_dem_routes_a_stim_like_circuit_through_detector_error_model():
    """a real stim.Circuit has flattened() AND num_detectors, so it must not be mistaken for a DEM --
    the reference's call (`decoder/base.py:151`) is circuit.detector_error_model(decompose_errors=False)."""
    from quits_amd.dem import as_dem
    text = helpers.circuit_text("bb72_custom_r6_p0.003")
    real = Circuit(text).detector_error_model()
    calls = []

    class FakeStimCircuit:                      # the attributes of stim.Circuit that fooled round 1's test order
        num_detectors = 288
        num_observables = 12

        def flattened(self):
            raise AssertionError("a circuit's flattened() is a circuit, not a DEM: must not be used")

        def detector_error_model(self, decompose_errors=True, **kw):
            calls.append(decompose_errors)
            return real

    assert as_dem(FakeStimCircuit()) is real and calls == [False]
    assert as_dem(real) is real                                # an actual DEM passes through
    assert as_dem(text).num_errors == real.num_errors          # plain text
    with pytest.raises(TypeError):
        as_dem(42)


def test_coordinates_are_ignored_and_noisy_measurements_rejected():
    """DETECTOR(x, y, t) / QUBIT_COORDS are valid Stim and carry no decoding information; M(p) is valid
    Stim measurement noise the QUITS emitter never produces -- refusing it beats silently dropping the noise."""
    plain = circuit_to_dem("R 0\nX_ERROR(0.125) 0\nM 0\nDETECTOR rec[-1]\n")
    coords = circuit_to_dem("QUBIT_COORDS(1, 2) 0\nR 0\nX_ERROR(0.125) 0\nM 0\nDETECTOR(1, 2, 0) rec[-1]\nSHIFT_COORDS(0, 0, 1)\n")
    assert coords.errors == plain.errors == [(0.125, (0,), ())]
    for op in ("M(0.01) 0", "MX(0.01) 0", "MR(0.01) 0"):
        with pytest.raises(NotImplementedError):
            flatten("R 0\n%s\nDETECTOR rec[-1]\n" % op)
    flatten("R 0\nM(0) 0\n")                                   # a zero argument is harmless


def test_all_detectors_circuit_shapes():
    """CircuitBuildOptions(get_all_detectors=True, noisy_zeroth_round=False, noisy_final_meas=True)
    (qldpc_code/circuit_construction/circuit_build_options.py:13-15): X and Z detectors together.  The [[72,12,6]] code has 36
    checks of each type; a Z-basis memory run of R = 2 rounds gives Z detectors in R + 2 blocks and X detectors in R blocks (the
    first X measurement of a |0> state is random, and the final data measurement is in the Z basis)."""
    dem = Circuit(helpers.circuit_text("bb72_custom_r2_alldet_p0.003")).detector_error_model()
    assert dem.num_observables == 12
    assert dem.num_detectors == 36 * (2 + 2) + 36 * 2, dem.num_detectors
    H, L, pri = detector_error_model_to_matrix(dem)
    assert H.shape[0] == dem.num_detectors and (np.diff(H.tocsc().indptr) > 0).all() and 0 < pri.min() and pri.max() < 0.5


@pytest.mark.parametrize("name,p_from,p_to", [("bb72_custom_r6_p0.003", 0.003, 0.0007), ("hgp225_cardinal_r3_p0.01", 0.01, 0.004),
                                              ("bb72_custom_r2_xbasis_mixed", None, None)])
def test_structure_cache_replays_the_same_numbers(monkeypatch, name, p_from, p_to):
    """The reference's notebooks call the decoder once per physical error rate with circuits that differ in the noise arguments
    only.  quits_amd.dem keeps the symptom structure per circuit structure and replays the probability folding for other
    arguments; decoder.base does the same for its merge by detector set.  Both must return the numbers of the full pass, bit for
    bit (same symptoms, same order, identical floats), and a circuit with another structure must not hit the cache."""
    from quits_amd import dem
    from quits_amd.decoder import base
    base_text = helpers.circuit_text(name)
    other = helpers.circuit_text_at_p(name, p_from, p_to) if p_from else base_text.replace("0.0030000000", "0.0012500000")
    assert other != base_text
    monkeypatch.setenv("QD_DEM_STRUCT_CACHE", "0")
    full = dem.circuit_to_dem(other)
    Hf, Lf, pf = base.detector_error_model_to_matrix(dem.Circuit(other))
    assert full.structure_key is None
    monkeypatch.setenv("QD_DEM_STRUCT_CACHE", "4")
    dem.dem_struct_cache_clear()
    base._MATRIX_CACHE.clear()
    first = dem.circuit_to_dem(base_text)
    base.detector_error_model_to_matrix(dem.Circuit(base_text))
    assert dem.dem_struct_cache_info()["misses"] == 1 and first.structure_key is not None
    replay = dem.circuit_to_dem(other)
    assert dem.dem_struct_cache_info()["hits"] >= 1 and replay.structure_key == first.structure_key
    assert replay.errors == full.errors                       # tuples of (float, detectors, observables): exact equality
    Hr, Lr, pr = base.detector_error_model_to_matrix(dem.Circuit(other))
    assert np.array_equal(pr, pf) and (Hr != Hf).nnz == 0 and (Lr != Lf).nnz == 0
    # a noise argument of zero drops mechanisms: another structure
    zeroed = base_text.replace("%.10f" % (p_from or 0.003), "%.10f" % 0.0)
    if zeroed != base_text:
        z = dem.circuit_to_dem(zeroed)
        assert z.structure_key != first.structure_key
    dem.dem_struct_cache_clear()
    base._MATRIX_CACHE.clear()


def test_structure_cache_keys_on_the_derived_probability_and_hands_out_copies(monkeypatch):
    """(1) a depolarising argument so small that the derived mechanism probability underflows to exactly 0.0 drops the
    mechanisms in the full pass, so such a circuit must not share a cached structure with the same circuit at an ordinary rate (in
    either direction); (2) detector_error_model_to_matrix is a public function: what it returns on a cache hit must be the caller's
    own copy; (3) QD_DEM_STRUCT_CACHE=0 switches the matrix cache off as well."""
    from quits_amd import dem
    from quits_amd.decoder import base
    name = "bb72_custom_r6_p0.003"
    text = helpers.circuit_text(name)
    tiny = text.replace("DEPOLARIZE1(0.0030000000)", "DEPOLARIZE1(1e-17)").replace("DEPOLARIZE2(0.0030000000)", "DEPOLARIZE2(1e-17)")
    assert tiny != text
    monkeypatch.setenv("QD_DEM_STRUCT_CACHE", "0")
    full_tiny, full_norm = dem.circuit_to_dem(tiny), dem.circuit_to_dem(text)
    assert len(full_tiny.errors) < len(full_norm.errors)            # the underflowed mechanisms are gone
    monkeypatch.setenv("QD_DEM_STRUCT_CACHE", "4")
    for first, second, want in ((tiny, text, full_norm), (text, tiny, full_tiny)):
        dem.dem_struct_cache_clear()
        base._MATRIX_CACHE.clear()
        a = dem.circuit_to_dem(first)
        b = dem.circuit_to_dem(second)
        assert a.structure_key != b.structure_key
        assert b.errors == want.errors
    # copies on a hit
    dem.dem_struct_cache_clear()
    base._MATRIX_CACHE.clear()
    H0, L0, p0 = base.detector_error_model_to_matrix(dem.Circuit(text))
    H1, L1, p1 = base.detector_error_model_to_matrix(dem.Circuit(text))
    assert H1 is not H0 and (H1 != H0).nnz == 0
    H1.data[:] = 0
    L1.data[:] = 0
    H2, L2, p2 = base.detector_error_model_to_matrix(dem.Circuit(text))
    assert (H2 != H0).nnz == 0 and (L2 != L0).nnz == 0 and np.array_equal(p2, p0)
    monkeypatch.setenv("QD_DEM_STRUCT_CACHE", "0")
    base._MATRIX_CACHE.clear()
    base.detector_error_model_to_matrix(dem.Circuit(text))
    assert len(base._MATRIX_CACHE) == 0


@pytest.mark.parametrize("name,max_row,min_row", [("hgp225_cardinal_r3_p0.01", 52, 21), ("bb72_custom_r6_p0.003", 35, 16),
                                                   ("bb144_custom_r12_p0.003", 35, 16)])
def test_row_weights_against_an_independent_forward_census(name, max_row, min_row):
    """SURVEY.md Appendix B quotes a maximum row weight of 53 for the HGP [[225,9,6]] R=3 DEM, the build's matrix has
    52 (same shape, non-zeros and sum of priors).  tests/dem_forward.py pushes every fault component of every noise instruction FORWARD
    through the circuit text (bit-packed Pauli frames, all components at once; nothing shared with the backward extractor but the
    parser) and counts the distinct detector sets: columns, non-zeros and the weight of EVERY row equal the build's -- 52 it is; the
    survey's throw-away probe mis-reported the digit.  (The QLP [[1020,136]] DEM, 3.2 M components, takes four minutes: run once,
    profiles/r06_dem_forward_census.txt -- rows 25..78 there, where Appendix B says 26..77.)"""
    import dem_forward
    from scipy.sparse import csr_matrix
    H, _, _ = helpers.dem_matrices(name)
    c = dem_forward.census(helpers.circuit_text(name))
    rw = np.diff(csr_matrix(H).indptr)
    assert c["columns"] == H.shape[1] and c["nnz"] == H.nnz
    assert np.array_equal(c["row_weights"], rw)
    assert np.array_equal(np.sort(c["col_weights"]), np.sort(np.diff(H.tocsc().indptr)))
    assert rw.max() == max_row and rw.min() == min_row
