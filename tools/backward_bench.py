#!/usr/bin/env python3
"""Timing of the sparse-conv backward kernels (row f-2, first part) on the bench scene: input gradient (the forward
kernel on the transposed map) and weight gradient (k_wgrad) per layer shape, algorithmic TFLOP/s = 2*pairs*Cin*Cout/t.
Usage: python tools/backward_bench.py [--voxels N] [--batch B]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agile3d_amd import backward as B  # noqa: E402
from agile3d_amd import lib as L  # noqa: E402
from agile3d_amd.engine import Scene  # noqa: E402
from agile3d_amd.synthetic import make_scene  # noqa: E402

CASES = [("L0 3^3  96-> 96", L.OP_CONV3, 0, 96, 96), ("L0 3^3 128-> 96", L.OP_CONV3, 0, 128, 96),
         ("L1 3^3  96-> 96", L.OP_CONV3, 1, 96, 96), ("L1 3^3  32-> 32", L.OP_CONV3, 1, 32, 32),
         ("L2 3^3 128->128", L.OP_CONV3, 2, 128, 128), ("L3 3^3 256->256", L.OP_CONV3, 3, 256, 256),
         ("L4 3^3 256->256", L.OP_CONV3, 4, 256, 256), ("L0 2^3s2 32->32", L.OP_DOWN, 0, 32, 32),
         ("L1 2^3tr 96->96", L.OP_UP, 1, 96, 96), ("L0 1x1  96->128", L.OP_LINEAR, 0, 96, 128)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=80_000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tape", action="store_true", help="also time a training-mode forward + backward of the backbone")
    ap.add_argument("--step", action="store_true", help="also time complete training iterations")
    ap.add_argument("--only", type=int, default=-1, help="only this case of the table (PMC passes: one shape per kernel name)")
    a = ap.parse_args()
    coords = np.concatenate([make_scene(a.voxels, seed=b, batch_index=b)["coords"] for b in range(a.batch)])
    sc = Scene(torch.from_numpy(coords).cuda())
    print("levels", sc.n)
    pairs3 = []
    for lvl in range(5):
        npad = (max(sc.n[lvl], 1) + 127) // 128 * 128
        nb = sc.table(lvl, L.TAB_NBR27).reshape(27, npad)
        pairs3.append(int((nb[:, :sc.n[lvl]] < sc.n[lvl]).sum()))
    for name, kind, lvl, cin, cout in (CASES if a.only < 0 else CASES[a.only:a.only + 1]):
        lo = B.level_out(kind, lvl)
        K = {L.OP_CONV3: 27, L.OP_DOWN: 8, L.OP_UP: 8, L.OP_LINEAR: 1}[kind]
        pairs = pairs3[lvl] if kind == L.OP_CONV3 else sc.n[min(lvl, lo)]
        x = torch.randn(sc.n[lvl], cin, device="cuda")
        dy = torch.randn(sc.n[lo], cout, device="cuda")
        w = torch.randn(K, cin, cout, device="cuda") / (cin * 4) ** 0.5
        t_w = timed(lambda: B.conv_weight_grad(sc, kind, lvl, x, dy), a.reps)
        t_x = timed(lambda: B.conv_input_grad(sc, kind, lvl, w, dy), a.reps)
        fl = 2.0 * pairs * cin * cout
        print(f"{name}: pairs {pairs:8d}  dW {1e3 * t_w:8.1f} us {fl / t_w / 1e9:6.1f} TF/s   "
              f"dx (incl. weight repack + buffer copies) {1e3 * t_x:8.1f} us {fl / t_x / 1e9:6.1f} TF/s")


def tape_time(voxels, batch):
    """Training-mode forward + backward of the whole backbone through BackboneTape (layer-at-a-time executor)."""
    import time
    from agile3d_amd import build_model, default_args
    from agile3d_amd.train_backbone import BackboneTape
    torch.manual_seed(0)
    model = build_model(default_args()).cuda().train()
    scenes = [make_scene(voxels, seed=b, batch_index=b) for b in range(batch)]
    coords = torch.from_numpy(np.concatenate([s["coords"] for s in scenes])).cuda()
    feats = torch.from_numpy(np.concatenate([s["feats"] for s in scenes])).cuda()
    sc = Scene(coords)
    d_out = torch.randn(sc.n[0], 128, device="cuda") / 16
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tape = BackboneTape(model, sc, feats)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        grads = tape.backward(d_out)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"BackboneTape, {batch} x {voxels} voxels: forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms "
          f"({len(grads)} gradients)")


def step_time(voxels, batch):
    """Complete training iterations (engine.py:38-150) on a synthetic labelled batch."""
    import random
    import time
    from agile3d_amd import batched_coordinates, build_model, default_args
    from agile3d_amd.criterion import build_mask_criterion
    from agile3d_amd.optim import AdamW
    from agile3d_amd.train_step import train_one_step
    torch.manual_seed(0)
    args = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])
    model = build_model(args).cuda()
    crit = build_mask_criterion(args)
    scenes = [make_scene(voxels, seed=b) for b in range(batch)]
    b = (batched_coordinates([s["coords"][:, 1:] for s in scenes]),
         torch.from_numpy(np.concatenate([s["raw_xyz"] for s in scenes])),
         torch.from_numpy(np.concatenate([s["feats"] for s in scenes])),
         [torch.from_numpy(s["labels"].astype(np.int64)) for s in scenes], None, None, [{} for _ in scenes],
         tuple(f"scene{i:04d}_00" for i in range(batch)), tuple(0 for _ in scenes))
    opt = AdamW(model.named_parameters(), lr=1e-4, weight_decay=1e-4)
    np.random.seed(1), random.seed(1)
    prof_it = int(os.environ.get("A3D_BB_CPROFILE", "-1"))     # host profile (cProfile) of this iteration, top 45 by cumulative time
    for it in range(int(os.environ.get("A3D_BB_ITERS", "4"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if it == prof_it:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            st = pr.runcall(train_one_step, model, crit, opt, b, torch.device("cuda"), 0.1)
            ps_ = pstats.Stats(pr).sort_stats("cumulative")
            ps_.print_stats(45)
            ps_.print_callers("torch.empty|method 'to' of|torch.tensor|torch.zeros")
        else:
            st = train_one_step(model, crit, opt, b, torch.device("cuda"), 0.1)
        torch.cuda.synchronize()
        ms = torch.cuda.memory_stats()
        print(f"training iteration {it}: {1e3 * (time.perf_counter() - t0):.0f} ms, loss {st['loss']:.3f}, clicks {st['clicks']}, "
              f"device allocations so far {ms.get('num_device_alloc', 0)}, frees {ms.get('num_device_free', 0)}, reserved "
              f"{ms.get('reserved_bytes.all.current', 0) >> 20} MB")


if __name__ == "__main__":
    main()
    if "--tape" in sys.argv:
        tape_time(80_000, 4)
    if "--step" in sys.argv:
        step_time(80_000, 4)
