"""world_size-2 coverage of the multi-GPU path on CPU (gloo): scene sharding, barrier-bracketed
timing with MAX over ranks, host-side gather.  The data path itself has no collective."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from agile3d_amd.sharding import gather_rows, scenes_of_rank, timed_steps
    from agile3d_amd.synthetic import make_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = scenes_of_rank(5, rank, world)
    scenes = [make_scene(1500, seed=i) for i in mine]
    delay = 0.02 * (rank + 1)   # rank 1 is slower: the reported time must be ITS time

    def step():
        time.sleep(delay)
        return sum(len(s["coords"]) for s in scenes)

    dt, out = timed_steps(step, 5, world, torch.device("cpu"))
    rows = gather_rows([(i, int(len(s["coords"]))) for i, s in zip(mine, scenes)], world)
    q.put((rank, mine, dt, out, sorted(rows)))
    dist.barrier()
    dist.destroy_process_group()


def test_scene_sharding_and_max_timing_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, dt0, _, rows0), (r1, s1, dt1, _, rows1) = res
    assert s0 == [0, 2, 4] and s1 == [1, 3]                 # disjoint, complete
    assert abs(dt0 - dt1) < 1e-9                            # every rank holds the MAX
    assert dt0 >= 5 * 0.04 * 0.95                           # ... which is the slow rank's time
    assert rows0 == rows1 and [r[0] for r in rows0] == [0, 1, 2, 3, 4]


def test_sharding_is_a_partition():
    from agile3d_amd.sharding import scenes_of_rank
    for n in (0, 1, 7, 64):
        for w in (1, 2, 3, 8):
            parts = [scenes_of_rank(n, r, w) for r in range(w)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from agile3d_amd.optim import allreduce_mean_
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    grads = {f"p{i}": torch.randn(shape, generator=g) for i, shape in enumerate([(27, 32, 32), (96,), (5, 7), (1, 128)])}
    before = {k: v.clone() for k, v in grads.items()}
    allreduce_mean_(grads, bucket_bytes=27 * 32 * 32 * 4 + 96 * 4)       # small buckets: several collectives
    q.put((rank, {k: v.numpy() for k, v in before.items()}, {k: v.numpy() for k, v in grads.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _overlap_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from agile3d_amd.optim import OverlappedAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(200 + rank)
    shapes = [(27, 32, 32), (96,), (5, 7), (1, 128), (8, 96, 96), (256,)]
    grads = {f"p{i}": torch.randn(shape, generator=g) for i, shape in enumerate(shapes)}
    before = {k: v.clone() for k, v in grads.items()}
    red = OverlappedAllReduce(bucket_bytes=27 * 32 * 32 * 4,             # several buckets in flight at once
                              expected={k: v.numel() for k, v in grads.items()})
    assert red.active and red.world == world and red.group is not None   # the buckets' own communicator
    try:
        red.add("nobody", torch.zeros(3))
        raise SystemExit("an unexpected gradient name must be refused")
    except RuntimeError:
        pass
    for k in ["p3", "p1", "p0"]:                                         # handed over as they "become final" ...
        red.add(k, grads[k])
    other = torch.randn(64, 64, generator=g) @ torch.randn(64, 64, generator=g)   # ... with work in between
    for k in ["p5", "p4", "p2"]:
        red.add(k, grads[k])
    out = red.finish({})
    assert sorted(out) == sorted(grads) and all(out[k] is grads[k] for k in grads) and other.shape == (64, 64)
    q.put((rank, {k: v.numpy() for k, v in before.items()}, {k: v.numpy() for k, v in grads.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _mismatch_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from agile3d_amd.optim import OverlappedAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    expected = {"a": 4, "b": 8} if rank == 0 else {"a": 4}               # rank 1 lacks one gradient
    try:
        OverlappedAllReduce(expected=expected)
        q.put((rank, "constructed"))
    except RuntimeError as e:
        q.put((rank, "refused: " + str(e)[:60]))
    # a missing hand-over is refused before anything waits on a collective
    red = OverlappedAllReduce(expected={"a": 4, "b": 8} if False else {"c": 2, "d": 2})
    red.add("c", torch.zeros(2))
    try:
        red.finish({})
        q.put((rank, "finished"))
    except RuntimeError as e:
        q.put((rank, "incomplete: " + str(e)[:40]))
    dist.barrier()
    dist.destroy_process_group()


def _reinit_worker(rank, world, ports, q):
    """destroy_process_group() + init_process_group() in ONE process (tests, notebooks): the reducer must not hand out the
    first world's bucket communicator, and a rank whose gradient list changes LATER (after lists that agreed) must be
    caught by the same check -- every rank runs the digest collective every time."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from agile3d_amd.optim import OverlappedAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    seen_groups = []
    for port in ports:
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        for it in range(2):                                             # twice per world: the second hits the cache
            g = {"a": torch.full((4,), float(rank + 1)), "b": torch.full((8,), float(10 * (rank + 1)))}
            red = OverlappedAllReduce(expected={k: v.numel() for k, v in g.items()})
            seen_groups.append(red.group)
            for k in g:
                red.add(k, g[k])
            red.finish({})
            assert torch.allclose(g["a"], torch.full((4,), 1.5)) and torch.allclose(g["b"], torch.full((8,), 15.0))
        assert seen_groups[-1] is seen_groups[-2]                       # one communicator per live world ...
        if len(seen_groups) > 2:
            assert seen_groups[-1] is not seen_groups[0]                # ... and not the destroyed world's
        # same world, lists that agreed so far; now rank 1's list changes: both ranks must get the error, nobody hangs
        try:
            OverlappedAllReduce(expected={"a": 4, "b": 8} if rank == 0 else {"a": 4, "b": 9})
            q.put((rank, "constructed"))
        except RuntimeError:
            q.put((rank, "refused"))
        dist.barrier()
        dist.destroy_process_group()
    assert len(OverlappedAllReduce._bucket_groups) <= 1                 # the first world's entry was dropped
    q.put((rank, "done"))


def test_overlapped_allreduce_survives_reinit_and_late_disagreement():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ports = [_free_port(), _free_port()]
    procs = [ctx.Process(target=_reinit_worker, args=(r, world, ports, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(3 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(m for _, m in got) == ["done"] * 2 + ["refused"] * 4, got


def test_overlapped_allreduce_refuses_ranks_that_disagree():
    """Ranks whose gradient lists differ fail with an error at construction (one small all-reduce of a digest per
    reducer) instead of hanging in a mis-sized collective; a gradient never handed over is reported by finish()."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(m.split(":")[0] for _, m in got) == ["incomplete", "incomplete", "refused", "refused"], got


def test_overlapped_allreduce_world2():
    """optim.OverlappedAllReduce (what train_one_step uses): gradients handed over one by one, asynchronous buckets, every
    rank ends with the mean in the tensors it handed over."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, a0), (_, b1, a1) = res
    for k in b0:
        want = (b0[k] + b1[k]) / 2
        assert a0[k].shape == b0[k].shape and np.allclose(a0[k], want, atol=1e-7) and np.array_equal(a0[k], a1[k])


def test_gradient_allreduce_mean_world2():
    """Training DP: every rank ends up with the mean of the ranks' gradients, tensor shapes untouched."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, a0), (_, b1, a1) = res
    for k in b0:
        want = (b0[k] + b1[k]) / 2
        assert a0[k].shape == b0[k].shape and np.allclose(a0[k], want, atol=1e-7) and np.array_equal(a0[k], a1[k])
