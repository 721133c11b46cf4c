"""Optimiser of the reference's training loop on the HIP library:

    clip_grad_norm_(grads, max_norm)      torch.nn.utils.clip_grad_norm_ (engine.py:145-148, max_norm = 0.1)
    AdamW(params, lr, weight_decay)       torch.optim.AdamW (main.py:125-127), .step(grads)
    allreduce_mean_(grads, group)         what DistributedDataParallel does to the gradients (one bucketed all-reduce,
                                          RCCL on the GPUs; SURVEY section 8e: 39.3 M fp32 = 157 MB per step)

Gradients are a dict {parameter name: tensor} as ``train_backbone.BackboneTape.backward`` returns them.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


MT_CHUNK = 4096   # A3D_MT_CHUNK
_MT_DTYPE = None


def _mt_dtype():
    global _MT_DTYPE
    import numpy as np
    if _MT_DTYPE is None:
        _MT_DTYPE = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("chunk0", "<i4"),
                              ("bias1", "<f4"), ("bias2_sqrt", "<f4"), ("pad", "<i4")])
        assert _MT_DTYPE.itemsize == 56
    return _MT_DTYPE


def _mt_layout(sizes):
    """The static half of a table of a3d_mt_tensor for tensors of ``sizes`` elements: (table with n / chunk0 filled in,
    number of chunks).  Built once per list of tensors (the 268 parameters of an iteration are the same every time)."""
    import numpy as np
    n = np.asarray(sizes, dtype=np.int64)
    chunks = (n + MT_CHUNK - 1) // MT_CHUNK
    tab = np.zeros(len(n), _mt_dtype())
    tab["n"] = n
    tab["chunk0"] = np.concatenate([[0], np.cumsum(chunks)[:-1]]).astype(np.int32)
    tab["bias1"] = 1.0
    tab["bias2_sqrt"] = 1.0
    return tab, int(chunks.sum())


def _to_device(tab, device):
    import numpy as np
    return torch.from_numpy(tab.view(np.uint8)).to(device)


_norm_layouts: dict = {}


def total_grad_norm(grads: dict) -> float:
    """sqrt(sum over all tensors of sum g^2): the 2-norm clip_grad_norm_ computes (fp64 accumulation on the device, all
    tensors in one launch + one ordered final sum).  Non-contiguous gradients are replaced IN ``grads`` by contiguous
    copies (the optimiser then reads those, it does not copy again)."""
    lib = L.load()
    gs = []
    f32 = torch.float32
    for k, g in grads.items():          # (one pass, the checks in the order that fails fastest: ~270 tensors per iteration)
        if g.dtype is not f32 or not g.is_cuda:
            raise RuntimeError("agile3d_amd.optim runs on the GPU only (fp32 CUDA tensors)")
        if not g.is_contiguous():
            g = grads[k] = g.contiguous()
        gs.append(g)
    key = tuple([g.numel() for g in gs])
    if 0 in key:
        gs = [g for g in gs if g.numel()]
        key = tuple([n for n in key if n])
    if not gs:
        return 0.0
    dev = gs[0].device
    lay = _norm_layouts.get(key)
    if lay is None:
        if len(_norm_layouts) > 8:
            _norm_layouts.clear()
        lay = _norm_layouts[key] = _mt_layout(key)
    tab, nchunks = lay[0].copy(), lay[1]
    tab["g"] = [g.data_ptr() for g in gs]
    tabd = _to_device(tab, dev)
    ws = torch.empty(lib.a3d_mt_workspace_bytes(nchunks), dtype=torch.uint8, device=dev)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    L.check(lib.a3d_sum_squares_multi(_ptr(tabd), len(gs), nchunks, _ptr(out), _ptr(ws), ws.numel(), _stream(gs[0])),
            "a3d_sum_squares_multi")
    return math.sqrt(float(out.item()))      # the one host synchronisation of the clip


def clip_grad_norm_(grads: dict, max_norm: float):
    """-> (total norm before clipping, coefficient to scale the gradients with); torch clamps max_norm / (norm + 1e-6)
    to 1.  The coefficient is applied inside ``AdamW.step`` (no extra pass over the gradients)."""
    norm = total_grad_norm(grads)
    coef = min(1.0, max_norm / (norm + 1e-6)) if max_norm > 0 else 1.0
    return norm, coef


# bumped by every AdamW.step(): the update kernel writes the parameters through raw pointers, which torch's tensor version
# counters do not see; caches of derived data (packed conv weights of the training tapes) key on it
WEIGHT_EPOCH = [0]


class AdamW:
    """torch.optim.AdamW's update rule, ONE kernel launch for all parameter tensors; state lives next to the parameters."""

    def __init__(self, named_params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = dict(named_params)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.state = {}
        self.steps = {}            # per parameter, like torch.optim.AdamW's state[p]['step']
        self.step_count = 0        # number of step() calls (= every parameter's step when all of them get gradients)
        self._layouts = {}         # names -> (static table, chunks, parameter pointers)

    def step(self, grads: dict, grad_scale: float = 1.0):
        """One launch for all tensors.  The table's static half (parameter / state pointers, sizes, chunk offsets) is kept
        per list of names -- the same every iteration --, only the gradient pointers and the bias corrections are filled
        in per call (the per-tensor Python of 268 entries was a third of the 3 ms this phase took)."""
        import numpy as np
        lib = L.load()
        self.step_count += 1
        WEIGHT_EPOCH[0] += 1
        names = tuple(k for k, g in grads.items() if self.params[k].numel())
        if not names:
            return
        lay = self._layouts.get(names)
        ptrs = tuple(self.params[k].data_ptr() for k in names)
        if lay is None or lay[2] != ptrs:
            for k in names:
                p = self.params[k]
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("AdamW: parameters must be contiguous fp32 CUDA tensors")
                if k not in self.state:
                    self.state[k] = (torch.zeros_like(p), torch.zeros_like(p))
            tab, nchunks = _mt_layout([self.params[k].numel() for k in names])
            tab["p"] = ptrs
            tab["m"] = [self.state[k][0].data_ptr() for k in names]
            tab["v"] = [self.state[k][1].data_ptr() for k in names]
            if len(self._layouts) > 4:
                self._layouts.clear()
            lay = self._layouts[names] = (tab, nchunks, ptrs)
        tab, nchunks = lay[0].copy(), lay[1]
        keep = []
        for k in names:
            g, p = grads[k], self.params[k]
            if g.shape != p.shape or not g.is_contiguous():
                g = g.reshape(p.shape).contiguous()
            keep.append(g)
        tab["g"] = [g.data_ptr() for g in keep]
        # bias corrections in double on the host, like torch's scalar path; every parameter counts ITS updates
        steps = self.steps
        t = np.array([steps.get(k, 0) + 1 for k in names], dtype=np.float64)
        steps.update(zip(names, t.astype(np.int64).tolist()))
        tab["bias1"] = 1.0 - self.betas[0] ** t
        tab["bias2_sqrt"] = np.sqrt(1.0 - self.betas[1] ** t)
        dev = self.params[names[0]].device
        tabd = _to_device(tab, dev)
        L.check(lib.a3d_adamw_step_multi(_ptr(tabd), len(names), nchunks, self.lr, self.betas[0], self.betas[1], self.eps,
                                         self.weight_decay, grad_scale, _stream(self.params[names[0]])), "a3d_adamw_step_multi")


def dist_all_reduce(t, group=None):
    """all_reduce(sum) in place; the gloo backend (CPU tests, and the two-ranks-on-one-GPU test) goes through host
    memory, nccl (= RCCL) works on the device tensor directly."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def dist_all_gather(t, group=None):
    """-> [world, ...] stack of every rank's ``t`` (same staging rule as dist_all_reduce)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    src = t.cpu() if (t.is_cuda and dist.get_backend(group) == "gloo") else t
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return torch.stack(parts).to(t.device)


def allreduce_mean_(grads: dict, group=None, bucket_bytes: int = 64 << 20):
    """Average the gradients over the data-parallel ranks in place: tensors are packed into ~64 MB buckets (few, large
    collectives: the xGMI links are bound per ring step) and all-reduced with torch.distributed
    (nccl = RCCL on the GPUs; gloo in the CPU tests)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grads
    world = dist.get_world_size(group)
    names = sorted(grads)
    i = 0
    while i < len(names):
        bucket, size = [], 0
        while i < len(names) and (not bucket or size + grads[names[i]].numel() * 4 <= bucket_bytes):
            bucket.append(names[i])
            size += grads[names[i]].numel() * 4
            i += 1
        flat = torch.cat([grads[n].reshape(-1) for n in bucket])
        dist_all_reduce(flat, group)
        flat /= world
        off = 0
        for n in bucket:
            k = grads[n].numel()
            grads[n].copy_(flat[off:off + k].view_as(grads[n]))
            off += k
    return grads


class OverlappedAllReduce:
    """Gradient averaging overlapped with the backward pass: gradients are handed over as they become final
    (``add``), packed into buckets of ``bucket_bytes`` and all-reduced asynchronously while the layers below are
    still being differentiated; ``finish`` waits, divides by the world size and writes the averages back.

    Order of completion in a training iteration: the decoder's parameters (all tapes done), then the U-Net from
    ``lin_squeeze_head`` / block8 down to the stem -- so the first buckets fly during the 20+ ms of backbone
    backward.  RCCL (backend nccl): the collective runs on the process group's own stream, which is ordered behind
    the stream that produced the bucket; gloo (CPU tests / two ranks on one GPU): the bucket is staged to host
    memory and reduced by gloo's thread.  Results do not depend on the bucket layout for two ranks (a sum of two
    numbers has one order); for more ranks a ring's order per element follows the layout, like any bucketed DDP."""

    _bucket_groups = {}      # id(parent group object) -> (parent group object, the communicator the buckets travel on)

    @classmethod
    def _bucket_group(cls, parent):
        """The buckets' own communicator for the LIVE process group ``parent`` (created once per group, collectively, with the
        parent's backend).  Entries are keyed on the group object itself -- a strong reference is kept, so an id cannot be
        reused while its entry exists -- and entries whose parent is no longer registered with torch.distributed (after
        ``destroy_process_group()`` + ``init_process_group()`` in one process: tests, notebooks) are dropped, so a stale
        communicator is never handed out."""
        import torch.distributed as dist
        try:
            live = dist.distributed_c10d._world.pg_map
            for k in [k for k, (pg, _) in cls._bucket_groups.items() if pg not in live]:
                del cls._bucket_groups[k]
        except Exception:                    # private API moved: fall back to "the default group changed -> forget everything"
            if any(pg is not parent for pg, _ in cls._bucket_groups.values()):
                cls._bucket_groups.clear()
        hit = cls._bucket_groups.get(id(parent))
        if hit is None or hit[0] is not parent:
            sub = dist.new_group(ranks=dist.get_process_group_ranks(parent), backend=dist.get_backend(parent))
            hit = cls._bucket_groups[id(parent)] = (parent, sub)
        return hit[1]

    def __init__(self, group=None, bucket_bytes: int = 32 << 20, expected=None, single_rank: bool = False):
        """``expected``: {name: numel} of every gradient this iteration will hand over (the same on every rank: the
        bucket layout follows the order of ``add`` calls).  A 63-bit digest of it is compared over the ranks by ONE small
        all-reduce per reducer -- every rank runs it every time, whatever it has seen before, so a rank whose list changed
        cannot enter a collective alone -- and ranks that disagree fail with an error instead of hanging inside a mis-sized
        bucket; ``add`` refuses unknown names and ``finish`` refuses to wait when one is missing.  ``single_rank``: run the
        collectives even in a world of one rank (exercises the RCCL path -- communicator, its stream, ordering -- on a
        single-GPU box)."""
        import torch.distributed as dist
        self.bucket_bytes = bucket_bytes
        up = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if up else 1
        self.active = up and (self.world > 1 or single_rank)
        self.parent = group
        self.group = group
        self.expected = dict(expected) if expected is not None else None
        self.seen = set()
        self.gloo = False
        self.pending, self.pending_bytes, self.flights = [], 0, []
        if self.active:
            parent = group if group is not None else dist.distributed_c10d._get_default_group()
            # the buckets get their own communicator: SyncBN's blocking all-reduces of the next iteration's forward (and
            # any other collective of the default group) then do not queue behind them on one RCCL stream
            self.group = OverlappedAllReduce._bucket_group(parent)
            self.gloo = dist.get_backend(self.group) == "gloo"      # staging follows the communicator the buckets really use
            if self.expected is not None:
                import hashlib
                h = int.from_bytes(hashlib.sha256(repr(sorted(self.expected.items())).encode()).digest()[:8], "big") >> 1
                dev = "cpu" if self.gloo else torch.device("cuda", torch.cuda.current_device())
                t = torch.tensor([h, -h], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)       # [max h, -min h]
                hi, neg_lo = (int(v) for v in t.tolist())
                if hi != h or -neg_lo != h:
                    raise RuntimeError("OverlappedAllReduce: the ranks disagree on the list of gradients (digest of this "
                                       f"rank {h:x}, over the ranks {-neg_lo:x}..{hi:x}); every rank must hand over the "
                                       "same tensors")

    def add(self, name, g):
        if not self.active:
            return
        if self.expected is not None:
            if name not in self.expected or self.expected[name] != g.numel():
                raise RuntimeError(f"OverlappedAllReduce.add: unexpected gradient {name!r} ({g.numel()} elements)")
            if name in self.seen:
                raise RuntimeError(f"OverlappedAllReduce.add: gradient {name!r} handed over twice")
        self.seen.add(name)
        self.pending.append((name, g))
        self.pending_bytes += g.numel() * 4
        if self.pending_bytes >= self.bucket_bytes:
            self.flush()

    def flush(self):
        import torch.distributed as dist
        if not self.active or not self.pending:
            return
        items, self.pending, self.pending_bytes = self.pending, [], 0
        flat = torch.cat([g.reshape(-1) for _, g in items])
        buf = flat.cpu() if (self.gloo and flat.is_cuda) else flat
        work = dist.all_reduce(buf, group=self.group, async_op=True)
        self.flights.append((items, flat, buf, work))

    def finish(self, grads: dict):
        """Wait for every bucket and write the averaged gradients into ``grads`` (in place where the tensor is there)."""
        if self.active and self.expected is not None and len(self.seen) != len(self.expected):
            missing = sorted(set(self.expected) - self.seen)
            raise RuntimeError(f"OverlappedAllReduce.finish: {len(missing)} expected gradients were never handed over "
                               f"(first: {missing[:3]}); the other ranks' buckets would not match")
        self.flush()
        for items, flat, buf, work in self.flights:
            work.wait()
            if buf is not flat:
                flat.copy_(buf)
            flat /= self.world
            off = 0
            for n, g in items:
                k = g.numel()
                g.copy_(flat[off:off + k].view_as(g))
                grads[n] = g
                off += k
        self.flights = []
        return grads


# ---- optimiser state in torch.optim.AdamW's state_dict() layout (so checkpoints interchange with the reference's)
def _adamw_state_dict(self):
    names = list(self.params)
    state = {}
    for i, n in enumerate(names):
        if n in self.state:
            state[i] = {"step": torch.tensor(float(self.steps.get(n, self.step_count))), "exp_avg": self.state[n][0].detach().cpu().clone(),
                        "exp_avg_sq": self.state[n][1].detach().cpu().clone()}
    group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
             "amsgrad": False, "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def _adamw_load_state_dict(self, sd):
    names = list(self.params)
    g = sd["param_groups"][0]
    self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
    self.state, self.steps = {}, {}
    self._layouts = {}             # the cached tables point at the old state tensors
    for i, st in sd["state"].items():
        p = self.params[names[int(i)]]
        self.state[names[int(i)]] = (st["exp_avg"].to(p.device, torch.float32).contiguous().clone(),
                                     st["exp_avg_sq"].to(p.device, torch.float32).contiguous().clone())
        self.steps[names[int(i)]] = int(float(st["step"]))
    self.step_count = max(self.steps.values()) if self.steps else 0


AdamW.state_dict = _adamw_state_dict
AdamW.load_state_dict = _adamw_load_state_dict
