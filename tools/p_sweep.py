#!/usr/bin/env python3
"""BASELINE.json configs[3]: the [[144,12,12]] window at p in {1..6}e-3, `--shots` shots per point, shots of every point
sharded over the ranks (one process per GPU), one 16-byte all-reduce per point -- the same code path as bench.py.

  python tools/p_sweep.py --shots 1000000                                       # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P tools/p_sweep.py --gpus 8
Prints one JSON line per point on rank 0."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--shots", type=int, default=1000000, help="shots per p-point, whole job")
    ap.add_argument("--batch", type=int, default=262144)
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--points", type=float, nargs="*", default=[0.001, 0.002, 0.003, 0.004, 0.005, 0.006])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dry-run-backend", default=None, help=argparse.SUPPRESS)   # tests: the per-point sharding + collectives on CPU (gloo), no decoding
    args = ap.parse_args()
    from quits_amd import parallel
    rank, world, local_rank = parallel.env_rank_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if args.dry_run_backend:
        # the N > 1 path without a GPU: join the group, then per point the barrier, the (errors, shots) SUM and the elapsed-time MAX
        dist = parallel.init_distributed(args.dry_run_backend)
        lo, hi = parallel.shard_range(args.shots, rank, world)
        for ip, p in enumerate(args.points):
            if dist is not None:
                dist.barrier()
            n_err, n_shots = parallel.reduce_counts(dist, (rank + 1) * (ip + 1), hi - lo)
            tmax = parallel.reduce_max(dist, float(rank + 1))
            if rank == 0:
                print(json.dumps({"dry_run": True, "p": p, "n_gpus": world, "errors": n_err, "shots": n_shots, "tmax": tmax}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("p_sweep.py needs a GPU; the decoder has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = parallel.init_distributed("nccl")
    import helpers
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.decoder.device import DemSampler, count_mismatch
    from quits_amd.decoder.sliding_window import build_circuit_plan
    from quits_amd.dem import Circuit
    hz = helpers.code("bb144")["hz"]
    R = 12
    lo, hi = parallel.shard_range(args.shots, rank, world)              # this rank's global shot indices, the same for every point
    for p in args.points:
        circ = Circuit(helpers.circuit_text("bb144_custom_r12_p%g" % p))
        H, Lobs, pri = detector_error_model_to_matrix(circ)
        opts = dict(bp_method="minimum_sum", schedule="parallel", max_iter=args.max_iter, osd_method="osd_0", osd_order=0)
        plan = build_circuit_plan(circ, hz, R + 2, 1, R, dict(opts), dict(opts))
        sampler = DemSampler(H, Lobs, pri)
        plan.decode(sampler.sample(min(args.batch, 4096), seed=args.seed, shot0=0)[0])          # warm-up
        fails = torch.zeros((1,), dtype=torch.int64, device="cuda")
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s0 in range(lo, hi, args.batch):
            nb = min(args.batch, hi - s0)
            det, obs = sampler.sample(nb, seed=args.seed, shot0=s0)      # counter-based sampler: a shard is a slice of the global stream
            fails += count_mismatch(plan.decode(det), obs)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = parallel.reduce_max(dist, time.perf_counter() - t0, "cuda")
        n_err, n_shots = parallel.reduce_counts(dist, int(fails.item()), hi - lo, "cuda")
        if rank == 0:
            pl = n_err / max(1, n_shots)
            print(json.dumps({"p": p, "shots": n_shots, "errors": n_err, "logical_error_rate": pl,
                              "ler_sigma": float(np.sqrt(max(pl * (1 - pl), 1e-30) / max(1, n_shots))), "n_gpus": world,
                              "shots_per_s_incl_sampling": n_shots / el, "seconds": el, "max_iter": args.max_iter,
                              "llr_grid_bits": plan.decoders()[0].info()["llr_grid_bits"]}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
