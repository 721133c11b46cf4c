"""Scene-sharded data parallelism for inference.

Scenes are independent in the reference's inference path (every batch sample is decoded on its own,
``models/agile3d.py:192``; BatchNorm uses running statistics), so N GPUs = N processes, rank r owns
scenes r, r+N, ... and there is NO collective on the data path.  ``torch.distributed`` (backend
``nccl`` = RCCL on ROCm, ``gloo`` in the CPU tests) is used only to line the ranks up and to reduce the
timing/metric scalars.
"""
from __future__ import annotations

import time

import torch


def scenes_of_rank(n_scenes: int, rank: int, world: int):
    """Static round-robin shard: scene i -> rank i % world."""
    return list(range(rank, n_scenes, world))


def timed_steps(step, steps: int, world: int, device, sync=None, own=None):
    """Time exactly `steps` calls of `step()` bracketed by barrier + device sync on both sides and
    return the MAX over ranks in seconds (the contract of bench.py).  `own` (a list) receives this
    rank's own time before the MAX."""
    import torch.distributed as dist
    # a process group at world size 1 (bench.py's RCCL check on a one-GPU box) goes through the same barriers / MAX
    lined_up = world > 1 or (dist.is_available() and dist.is_initialized())
    sync = sync or (torch.cuda.synchronize if device.type == "cuda" else (lambda: None))
    sync()
    if lined_up:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    if lined_up:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if own is not None:
        own.append(dt)
    if lined_up:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def gather_rows(rows, world: int):
    """Host-side gather of small per-scene result rows (IoU, timings) to every rank."""
    if world == 1:
        return list(rows)
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, list(rows))
    return [r for part in out for r in part]
