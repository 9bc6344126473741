"""World-size-2 run of the multi-GPU plumbing on CPU (gloo): shot sharding + the one (errors, shots) all-reduce."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import helpers, oracle as orc
from quits_amd import parallel
rank, world, _ = parallel.env_rank_world()
dist = parallel.init_distributed("gloo")
assert (dist is not None) == (world > 1)
N = 301
lo, hi = parallel.shard_range(N, rank, world)
H, L, pri = helpers.dem_matrices("bb72_custom_r6_p0.003")
synd, obs, _ = orc.sample_dem(H, L, pri, seed=42, shot0=lo, B=hi - lo)         # counter-based sampler: shard = slice
err, _ = orc.Graph(H, pri).decode_batch(synd, orc.make_params("minimum_sum", "parallel", 10, "osd_0", 0, 1.0, orc.FORM_LDPC_F64))
fails = int(((np.asarray(L @ err.T %% 2).T != obs).any(axis=1)).sum())
tot_err, tot_shots = parallel.reduce_counts(dist, fails, hi - lo)
tmax = parallel.reduce_max(dist, float(rank + 1))
if rank == 0:
    print(json.dumps({"errors": tot_err, "shots": tot_shots, "tmax": tmax, "local": [lo, hi]}))
if dist is not None:
    dist.barrier(); dist.destroy_process_group()
'''


def _run(world):
    code = WORKER % {"root": ROOT}
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    if world == 1:
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    else:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                              "--master-addr", "127.0.0.1", "--master-port", "29617", _script(code)],
                             env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def _script(code):
    import tempfile
    f = tempfile.NamedTemporaryFile("w", suffix="_worker.py", delete=False)
    f.write(code)
    f.close()
    return f.name


def test_shard_range_partitions():
    from quits_amd.parallel import shard_range
    for total in (0, 1, 7, 301, 10 ** 6):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_equal_one_rank():
    one = _run(1)
    two = _run(2)
    assert one["shots"] == two["shots"] == 301
    assert one["errors"] == two["errors"]           # same global shot indices -> same syndromes -> same failures
    assert two["tmax"] == 2.0 and one["tmax"] == 1.0


def _torchrun(world, script_args, port):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("MASTER_PORT", None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args,
                          env=env, capture_output=True, text=True, timeout=600)


def test_bench_rank_plumbing_two_processes():
    """bench.py's N > 1 branch as the driver launches it (torch.distributed.run, one rank per GPU), up to the first CUDA call:
    RANK / WORLD_SIZE / MASTER_* from the environment, the --gpus check, group start-up, barrier, the (errors, shots) SUM and the
    elapsed-time MAX -- on gloo, because there is no GPU here."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    out = _torchrun(2, [bench, "--gpus", "2", "--shots", "1001", "--dry-run-backend", "gloo"], 29641)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"dry_run": True, "n_gpus": 2, "errors": 3, "shots": 2002, "tmax": 2.0}
    # BASELINE configs[4] is an 8-GPU configuration: the QLP sliding-window command line takes the same branch
    qlp = _torchrun(2, [bench, "--gpus", "2", "--dry-run-backend", "gloo", "--code", "qlp1020", "--window", "3", "1", "--shots", "8192",
                        "--p-override", "0.001", "--osd-method", "osd_cs", "--osd-order", "1"], 29647)
    assert qlp.returncode == 0, qlp.stderr[-2000:]
    line = json.loads([ln for ln in qlp.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"dry_run": True, "n_gpus": 2, "errors": 3, "shots": 16384, "tmax": 2.0}
    bad = _torchrun(2, [bench, "--gpus", "1", "--dry-run-backend", "gloo"], 29643)
    assert bad.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in (bad.stderr + bad.stdout)
    # without --dry-run-backend the next thing bench.py does is ask for a GPU, and says so
    nogpu = _torchrun(2, [bench, "--gpus", "2"], 29645)
    import torch
    if not torch.cuda.is_available():
        assert nogpu.returncode != 0 and "needs a GPU" in (nogpu.stderr + nogpu.stdout)


def test_p_sweep_two_processes_six_points():
    """tools/p_sweep.py (BASELINE configs[3]) as it would be launched on N GPUs, on gloo: six p-points, the same shard of the
    global shot range for every point, one (errors, shots) all-reduce and one elapsed-time MAX per point."""
    import json
    script = os.path.join(ROOT, "tools", "p_sweep.py")
    out = _torchrun(2, [script, "--gpus", "2", "--shots", "1000001", "--dry-run-backend", "gloo"], 29651)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert [ln["p"] for ln in lines] == [0.001, 0.002, 0.003, 0.004, 0.005, 0.006]
    for ip, ln in enumerate(lines):
        assert ln["shots"] == 1000001 and ln["errors"] == 3 * (ip + 1) and ln["tmax"] == 2.0 and ln["n_gpus"] == 2
    bad = _torchrun(2, [script, "--gpus", "8", "--dry-run-backend", "gloo"], 29653)
    assert bad.returncode != 0 and "--gpus 8 but WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def test_master_port_comes_from_the_launcher():
    from quits_amd import parallel
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    env.pop("MASTER_PORT", None)
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from quits_amd import parallel; parallel.init_distributed('gloo')" % ROOT],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "MASTER_PORT is not set" in out.stderr
