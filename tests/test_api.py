"""The C-ABI library: loads without a GPU, exports every symbol include/quits_amd.h declares; package surface."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "quits_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from quits_amd import _lib
    L = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), "libquits_amd.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms            # the Python binding list is the header's list
    assert L.qd_version() >= 100
    assert isinstance(L.qd_last_error(), bytes)


def test_no_compute_without_gpu_but_argument_checks_work():
    import torch
    from quits_amd import _lib
    L = _lib.load()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert L.qd_device_count() == 0
    h = ctypes.c_void_p()
    rc = L.qd_graph_create(0, 0, None, None, None, 0, ctypes.byref(h))
    assert rc == -1 and b"empty" in L.qd_last_error()
    with pytest.raises(RuntimeError, match="no HIP device"):
        _lib.require_gpu()
    assert L.qd_decoder_post_head_start(None, 50, None) == -1 and b"null decoder" in L.qd_last_error()      # (C-ABI 103: argument check, no launch)


def test_package_surface_mirrors_reference():
    """Names of quits.decoder.__all__ (/root/reference/src/quits/decoder/__init__.py:13-24)."""
    import inspect
    import quits_amd.decoder as d
    for name in ("dict_to_csc_matrix_column_row", "dict_to_csc_matrix_row_column", "detector_error_model_to_matrix",
                 "spacetime", "sliding_window_phenom_mem", "sliding_window_circuit_mem",
                 "sliding_window_bposd_phenom_mem", "sliding_window_bposd_circuit_mem",
                 "sliding_window_bplsd_phenom_mem", "sliding_window_bplsd_circuit_mem"):
        assert name in d.__all__ and callable(getattr(d, name))
    sig = inspect.signature(d.sliding_window_bposd_circuit_mem)
    assert list(sig.parameters) == ["zcheck_samples", "circuit", "hz", "lz", "W", "F", "max_iter", "osd_order",
                                    "bp_method", "schedule", "osd_method", "tqdm_on"]
    assert [sig.parameters[k].default for k in ("max_iter", "osd_order", "bp_method", "schedule", "osd_method", "tqdm_on")] \
        == [2, 0, "product_sum", "serial", "osd_cs", False]
    sig = inspect.signature(d.sliding_window_bposd_phenom_mem)
    assert list(sig.parameters) == ["zcheck_samples", "hz", "lz", "W", "F", "eff_error_rate_per_fault", "max_iter",
                                    "osd_order", "bp_method", "schedule", "osd_method", "tqdm_on", "error_rate"]
    sig = inspect.signature(d.sliding_window_circuit_mem)
    assert list(sig.parameters) == ["zcheck_samples", "circuit", "hz", "lz", "W", "F", "decoder1", "decoder2", "dict1",
                                    "dict2", "error_rate_name1", "error_rate_name2", "function_name1", "function_name2",
                                    "tqdm_on"]
    # BP-LSD (reference bplsd.py:10,54): same names, positional order, keyword names and defaults
    sig = inspect.signature(d.sliding_window_bplsd_circuit_mem)
    assert list(sig.parameters) == ["zcheck_samples", "circuit", "hz", "lz", "W", "F", "max_iter", "lsd_order",
                                    "bp_method", "schedule", "lsd_method", "tqdm_on"]
    assert [sig.parameters[k].default for k in ("max_iter", "lsd_order", "bp_method", "schedule", "lsd_method", "tqdm_on")] \
        == [2, 0, "product_sum", "serial", "lsd_cs", False]
    sig = inspect.signature(d.sliding_window_bplsd_phenom_mem)
    assert list(sig.parameters) == ["zcheck_samples", "hz", "lz", "W", "F", "eff_error_rate_per_fault", "max_iter",
                                    "lsd_order", "bp_method", "schedule", "lsd_method", "tqdm_on", "error_rate"]
    assert issubclass(d.BpLsdDecoder, d.BpOsdDecoder)


def test_bplsd_options_outside_the_device_path_fail_loudly():
    """bits_per_step != 1 and orders beyond the kernels' limits are legal for ldpc and not implemented on the device:
    NotImplementedError, raised before anything touches the GPU; a missing error rate is the reference's ValueError
    (bplsd.py:35-36).  lsd_order = 1 -- what the reference's own calls pass -- maps onto the device decoder."""
    import helpers
    from quits_amd.decoder import sliding_window_bplsd_circuit_mem, sliding_window_bplsd_phenom_mem
    from quits_amd.decoder.bplsd import lsd_to_device_method
    from quits_amd.dem import Circuit
    cd = helpers.code("bb72")
    det = np.zeros((4, 36 * 8), np.uint8)
    with pytest.raises(NotImplementedError, match="lsd_order"):
        sliding_window_bplsd_circuit_mem(det, Circuit(helpers.circuit_text("bb72_custom_r6_p0.003")), cd["hz"], cd["lz"], 3, 1, lsd_order=65)
    with pytest.raises(NotImplementedError, match="lsd_order"):
        lsd_to_device_method("lsd_e", 16)
    with pytest.raises(ValueError, match="eff_error_rate_per_fault"):
        sliding_window_bplsd_phenom_mem(det, cd["hz"], cd["lz"], 3, 1)
    assert lsd_to_device_method("lsd_cs", 0) == lsd_to_device_method("lsd_e", 0) == lsd_to_device_method("lsd_0", 0) == ("lsd_0", 0)
    assert lsd_to_device_method("lsd_cs", 1) == ("lsd_cs", 1) and lsd_to_device_method("LSD_E", 7) == ("lsd_e", 7)
    assert lsd_to_device_method("lsd_0", 3) == ("lsd_0", 0)
    with pytest.raises(NotImplementedError):
        lsd_to_device_method("lsd_0", 0, bits_per_step=2)
    with pytest.raises(ValueError):
        lsd_to_device_method("osd_cs", 0)


def test_device_options_follow_the_decoder_class():
    """the post-processor is chosen by the plug-in CLASS, not by which keys the dict happens to hold."""
    from quits_amd.decoder import BpLsdDecoder, BpOsdDecoder
    from quits_amd.decoder.sliding_window import _kwargs_for_device
    base = {"bp_method": "product_sum", "max_iter": 3, "schedule": "serial", "channel_probs": None}
    kw = _kwargs_for_device(base, BpLsdDecoder)                      # no lsd_* keys: still LSD (ldpc's defaults lsd_0 / order 0)
    assert kw["osd_method"] == "lsd_0" and kw["osd_order"] == 0
    kw = _kwargs_for_device(dict(base, osd_method="osd_cs", osd_order=2), BpOsdDecoder)
    assert kw["osd_method"] == "osd_cs" and kw["osd_order"] == 2
    assert "osd_method" not in _kwargs_for_device(base, BpOsdDecoder)            # BatchDecoder's default then applies
    with pytest.raises(TypeError, match="lsd_order"):
        _kwargs_for_device(dict(base, lsd_order=0), BpOsdDecoder)
    with pytest.raises(TypeError, match="osd_method"):
        _kwargs_for_device(dict(base, osd_method="osd_0"), BpLsdDecoder)

    class MyLsd(BpLsdDecoder):
        pass
    assert _kwargs_for_device(base, MyLsd)["osd_method"] == "lsd_0"
    # legal ldpc keywords that change nothing here are dropped, the ones the device path lacks say so
    kw = _kwargs_for_device(dict(base, omp_thread_count=4, input_vector_type="syndrome", random_schedule_seed=None,
                                 serial_schedule_order=None), BpOsdDecoder)
    assert set(kw) == {"bp_method", "max_iter", "schedule"}
    with pytest.raises(NotImplementedError, match="natural fault order"):
        _kwargs_for_device(dict(base, random_schedule_seed=7), BpOsdDecoder)
    with pytest.raises(NotImplementedError, match="syndromes only"):
        _kwargs_for_device(dict(base, input_vector_type="received_vector"), BpLsdDecoder)


def test_plan_cache_key_and_lru(monkeypatch):
    """plans are cached on everything they depend on -- circuit text, hz, W, F, rounds, plug-in classes, option
    dicts (arrays by content) -- and evicted least recently used first."""
    from quits_amd.decoder import BpLsdDecoder, BpOsdDecoder
    from quits_amd.decoder import sliding_window as sw
    hz = np.eye(3, dtype=int)
    d = {"bp_method": "minimum_sum", "max_iter": 5, "channel_probs": np.array([0.1, 0.2])}
    k0 = sw.plan_key("circuit", "H 0\nM 0\n", hz, None, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, d, d)
    assert k0 == sw.plan_key("circuit", "H 0\nM 0\n", hz.copy(), None, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, dict(d), {**d, "channel_probs": np.array([0.1, 0.2])})
    hash(k0)
    others = [sw.plan_key("circuit", "H 0\nM 1\n", hz, None, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, d, d),
              sw.plan_key("circuit", "H 0\nM 0\n", 1 - hz, None, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, d, d),
              sw.plan_key("circuit", "H 0\nM 0\n", hz, None, 5, 3, 6, BpOsdDecoder, BpOsdDecoder, d, d),
              sw.plan_key("circuit", "H 0\nM 0\n", hz, None, 3, 1, 7, BpOsdDecoder, BpOsdDecoder, d, d),
              sw.plan_key("circuit", "H 0\nM 0\n", hz, None, 3, 1, 6, BpOsdDecoder, BpLsdDecoder, d, d),
              sw.plan_key("circuit", "H 0\nM 0\n", hz, None, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, d, {**d, "max_iter": 6}),
              sw.plan_key("circuit", "H 0\nM 0\n", hz, None, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, d, {**d, "channel_probs": np.array([0.1, 0.3])}),
              sw.plan_key("phenom", None, hz, hz, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, d, d)]
    assert len(set(others + [k0])) == len(others) + 1
    sw.plan_cache_clear()
    monkeypatch.setenv("QD_PLAN_CACHE", "2")
    built = []
    for key in ("a", "b", "a", "c", "b"):
        sw.cached_plan(key, lambda key=key: built.append(key) or key.upper())
    assert built == ["a", "b", "c", "b"]                 # 'a' hit once; 'b' was evicted when 'c' came in
    assert sw.plan_cache_info() == {"size": 2, "capacity": 2, "hits": 1, "misses": 4}
    monkeypatch.setenv("QD_PLAN_CACHE", "0")
    assert sw.cached_plan("a", lambda: "fresh") == "fresh"
    sw.plan_cache_clear()


def test_dict_helpers():
    from quits_amd.decoder import dict_to_csc_matrix_column_row, dict_to_csc_matrix_row_column
    a = dict_to_csc_matrix_column_row({0: [1, 2], 2: [0]}, (3, 3))
    assert a.toarray().tolist() == [[0, 0, 1], [1, 0, 0], [1, 0, 0]]
    b = dict_to_csc_matrix_row_column({frozenset([0, 2]): 1, frozenset([1]): 0}, (3, 2))
    assert b.toarray().tolist() == [[0, 1], [1, 0], [0, 1]]
    assert a.dtype == np.uint8


def test_product_path_does_not_import_the_oracle():
    """quits_amd/ must never reach into oracle/ (the CPU restatement is test infrastructure)."""
    pkg = os.path.join(ROOT, "quits_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "liboracle" not in txt and "qd_oracle" not in txt.replace("oracle/qd_oracle.c", ""), f


def _resource_usage(tmp_path, src):
    """(kernel name, ScratchSize bytes per lane) for every kernel hipcc compiles from quits_amd/csrc/<src> with the Makefile's flags."""
    import re
    import subprocess
    cs = os.path.join(ROOT, "quits_amd", "csrc")
    mk = open(os.path.join(cs, "Makefile")).read()
    flags = re.search(r"^FLAGS := (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    flags = [f for f in flags if f != "-shared"]
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", "-o",
                          str(tmp_path / (src + ".o")), os.path.join(cs, src)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    assert len(names) == len(scratch)
    return list(zip(names, scratch))


# every kernel a default plan can launch -- and every fall-back behind it -- stays out of scratch.  (file, kernel-name filter, how many
# instantiations the filter must find, why.)  The full-rank row form qd_osd0_reg_kernel<., ., true> (OSD-CS / OSD-E on windows of more than
# 1408 detectors, which no BASELINE config has) is the one kernel left that spills, and the test says so by excluding it by name.
_NO_SCRATCH = [
    ("osd_sr.hip", "qd_osd0_sr_kernel", 8, "keeps ~100 spilled scalar registers in lanes of vector registers: only safe while no vector register "
                                           "is spilled (a spill-heavy build faulted on the GPU, DESIGN.md K2s)"),
    ("osd_cs.hip", "qd_osdcs_kernel", 3, "OSD-CS / OSD-E, the reference wrapper's default post-processor (bposd.py:54); the kernel it replaced spilled 648-684 B"),
    ("bp_scatter_wide.hip", "qd_bp_scatter_wide_kernel", 6, "flooding min-sum, the headline's BP kernel"),
    ("bp_scatter.hip", "qd_bp_scatter_kernel", 3, "flooding min-sum, one check per lane"),
    ("bp_kernels.hip", "qd_bp_minsum_kernel", 36, "the recheck / coarse-grid pass behind the scatter kernels (VERDICT r5 weak 9: 96 instantiations with 12-16 B each)"),
    ("osd_kernels.hip", "Lb0E", 8, "qd_osd0_reg_kernel<., ., false>: the shots qd_osd0_sr_kernel hands over (VERDICT r5 weak 9: 88-144 B)"),
    ("gf2_kernels.hip", "qd_", 7, "acc ^= L e, U e, the sampler, unpack, mismatch count, the hold of the two-stream driver"),
    ("lsd_kernels.hip", "qd_lsd", 4, "BP-LSD"),
]


@pytest.mark.parametrize("src,pattern,count,why", _NO_SCRATCH, ids=[x[0] + ":" + x[1] for x in _NO_SCRATCH])
def test_kernels_use_no_scratch(tmp_path, src, pattern, count, why):
    kern = [(n, sc) for n, sc in _resource_usage(tmp_path, src) if pattern in n]
    assert len(kern) == count, (len(kern), [n for n, _ in kern])
    assert all(sc == 0 for _, sc in kern), [k for k in kern if k[1]]


def test_serial_edge_kernel_keeps_five_workgroups_per_cu(tmp_path):
    """qd_bp_edge_kernel, serial schedule, column weight <= 8 (the reference wrapper's default BP on every BASELINE window): at most 96 vector
    registers and no scratch, so that five workgroups of four wavefronts share a CU.  97 registers = four workgroups: 55 -> 76 ms per launch on
    the W = 5 windows (profiles/r06_k1g_staged_ab.txt) -- a shot index held across the sweeps was enough."""
    import re
    import subprocess
    cs = os.path.join(ROOT, "quits_amd", "csrc")
    mk = open(os.path.join(cs, "Makefile")).read()
    flags = [f for f in re.search(r"^FLAGS := (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split() if f != "-shared"]
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", "-o",
                          str(tmp_path / "g.o"), os.path.join(cs, "bp_general.hip")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    vgprs = [int(x) for x in re.findall(r"VGPRs: (\d+)", out.stderr)]
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    assert len(names) == len(vgprs) == len(scratch)
    serial = [(n, v, sc) for n, v, sc in zip(names, vgprs, scratch) if re.search(r"qd_bp_edge_kernelILi[01]ELi1ELi4ELi[48]E", n)]
    assert len(serial) == 8, [n for n, _, _ in serial]          # 2 methods x column weight 4 / 8 x prefixes in LDS or not
    assert all(v <= 96 and sc == 0 for _, v, sc in serial), serial


def test_lane_groups_of_the_pipelined_driver():
    """Chunks are dealt to the driver's lanes in as few groups as possible, as even as they come, in order."""
    from quits_amd.decoder.sliding_window import lane_groups
    assert lane_groups(0, 3) == [] and lane_groups(1, 2) == [1] and lane_groups(2, 3) == [2] and lane_groups(3, 3) == [3]
    assert lane_groups(4, 3) == [2, 2] and lane_groups(5, 3) == [3, 2] and lane_groups(7, 3) == [3, 2, 2] and lane_groups(16, 3) == [3, 3, 3, 3, 2, 2]
    assert lane_groups(16, 2) == [2] * 8 and lane_groups(3, 2) == [2, 1]
    for n in range(1, 40):
        for lanes in (2, 3, 4):
            g = lane_groups(n, lanes)
            assert sum(g) == n and max(g) <= lanes and max(g) - min(g) <= 1 and len(g) == -(-n // lanes) and g == sorted(g, reverse=True)


def test_lanes_and_chunk_give_way_to_the_workspace_budget():
    """Plans of many large windows do not get three lanes of 65 536 shots: the third lane goes first, then the chunk is halved."""
    from quits_amd.decoder.sliding_window import fit_lanes_and_chunk
    GB = 1 << 30
    assert fit_lanes_and_chunk(12 * 8704, 3, 65536, 160 * GB) == (3, 65536)                 # BB144 W = 3 / F = 1: 20 GB
    qlp = 18 * (4 * 18944 + 64)                                                            # QLP [[1020,136]] W = 3: 1.36 MB per shot and lane
    assert fit_lanes_and_chunk(qlp, 3, 65536, 160 * GB) == (2, 32768)
    assert fit_lanes_and_chunk(qlp, 2, 65536, 288 * GB) == (2, 65536)
    assert fit_lanes_and_chunk(10 ** 9, 3, 65536, GB) == (2, 8192)                          # never below the floor, never fewer than two lanes


def test_plan_cache_is_process_wide_locked_and_keyed_on_the_device(monkeypatch):
    """a cached plan is bound to the device it was built on -- the key holds the current device -- and carries
    mutable state (staging buffers, side streams, decoder workspaces): ONE cache per process under a module lock, use serialised by
    the plan's own re-entrant lock; a lookup releases the workspaces of every other cached plan that is idle, whatever thread used it
    last, and leaves alone a plan another thread holds."""
    import threading
    from quits_amd.decoder import BpOsdDecoder
    from quits_amd.decoder import sliding_window as sw
    hz = np.eye(3, dtype=int)
    d = {"bp_method": "minimum_sum", "max_iter": 5}
    k0 = sw.plan_key("circuit", "H 0\nM 0\n", hz, None, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, d, d)
    assert ("device", sw._current_device()) in k0
    monkeypatch.setattr(sw, "_current_device", lambda: 5)
    k5 = sw.plan_key("circuit", "H 0\nM 0\n", hz, None, 3, 1, 6, BpOsdDecoder, BpOsdDecoder, d, d)
    assert k5 != k0 and ("device", 5) in k5

    class FakePlan:
        def __init__(self, name):
            self.name, self._lock, self.released = name, threading.RLock(), 0

        def release_workspaces(self):
            self.released += 1

    sw.plan_cache_clear()
    px = sw.cached_plan("x", lambda: FakePlan("x"))
    seen = {}

    def other():
        seen["same"] = sw.cached_plan("x", lambda: FakePlan("worker's x"))            # shared, not rebuilt
        seen["info"] = sw.plan_cache_info()
        seen["y"] = sw.cached_plan("y", lambda: FakePlan("y"))                       # idle x loses its workspaces

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen["same"] is px and seen["info"]["hits"] == 1 and seen["info"]["misses"] == 1 and seen["info"]["size"] == 1
    assert px.released == 1 and px._ws_live is False and seen["y"].released == 0
    # a plan that is in use in another thread keeps its workspaces
    assert sw.cached_plan("x", lambda: None) is px and seen["y"].released == 1
    held, go = threading.Event(), threading.Event()

    def holder():
        with px._lock:
            held.set()
            go.wait(10)

    t = threading.Thread(target=holder)
    t.start()
    held.wait(10)
    sw.cached_plan("y", lambda: None)
    assert px.released == 1                    # busy: left alone
    go.set()
    t.join()
    sw.cached_plan("y", lambda: None)
    assert px.released == 2
    # concurrent misses on one key build once
    built = []
    def build():
        built.append(1)
        return FakePlan("z")
    ts = [threading.Thread(target=lambda: sw.cached_plan("z", build)) for _ in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(built) == 1
    sw.plan_cache_clear()
    assert sw.plan_cache_info() == {"size": 0, "capacity": 8, "hits": 0, "misses": 0}


def test_env_switches_parse_loosely(monkeypatch):
    """QD_NO_PIPELINE=true / yes still switch the pipelined driver off; a malformed QD_HOST_PIECE_SHOTS falls back."""
    from quits_amd.decoder import sliding_window as sw
    for v, want in (("1", True), ("true", True), ("yes", True), ("0", False), ("", False), ("off", False)):
        monkeypatch.setenv("QD_NO_PIPELINE", v)
        assert sw._env_flag("QD_NO_PIPELINE") is want, v
    monkeypatch.setenv("QD_HOST_PIECE_SHOTS", "a lot")
    assert sw._env_int("QD_HOST_PIECE_SHOTS", 123) == 123
