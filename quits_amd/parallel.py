"""Multi-GPU plumbing: one process per GPU, shots sharded contiguously, one 16-byte all-reduce at the end.

The decoding path has no exchange step (shots are independent: the reference's loop body only reads row i,
quits/decoder/sliding_window.py:162), so the only collective is the SUM of (logical errors, shots) -- RCCL when the
backend is "nccl" on ROCm, gloo in the CPU tests."""
from __future__ import annotations

import os
from typing import Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` shots owned by `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(total), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend: str = "nccl"):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (torch.distributed.run sets them).
    Returns torch.distributed, or None for a single process."""
    rank, world, _ = env_rank_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")       # the container's hostname may not resolve
    if "MASTER_PORT" not in os.environ:                     # the launcher (torch.distributed.run) owns the port
        raise RuntimeError("WORLD_SIZE=%d but MASTER_PORT is not set: launch with `python -m torch.distributed.run --nnodes=1 "
                           "--nproc-per-node N --master-addr 127.0.0.1 --master-port P ...`" % world)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def _collective_device(dist, device):
    """gloo reduces host tensors whatever the caller asked for; RCCL ("nccl") needs them on the rank's GPU."""
    return "cpu" if dist.get_backend() == "gloo" else device


def reduce_counts(dist, n_errors: int, n_shots: int, device="cpu") -> Tuple[int, int]:
    """All-reduce SUM of the two counters; identity without a process group."""
    if dist is None:
        return int(n_errors), int(n_shots)
    import torch
    device = _collective_device(dist, device)
    t = torch.tensor([int(n_errors), int(n_shots)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t[0].item()), int(t[1].item())


def reduce_max(dist, value: float, device="cpu") -> float:
    if dist is None:
        return float(value)
    import torch
    device = _collective_device(dist, device)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
