#!/usr/bin/env python3
"""Fit the model on a few seeded synthetic labelled scenes (agile3d_amd.fit) and print the interactive protocol's
IoU@k / NoC table of the GPU product every --every iterations: how bench.py's `iou_at_k` recipe was chosen.
    python tools/fit_synthetic.py --scenes 4 --voxels 5000 --iters 300 --lr 1e-3 --every 100"""
import argparse
import contextlib
import io
import json
import os
import random
import sys
import tempfile
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--voxels", type=int, default=5000)
    ap.add_argument("--objects", type=int, default=3)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--colour", type=float, default=0.0)
    ap.add_argument("--max-clicks", type=int, default=20)
    a = ap.parse_args()
    from agile3d_amd import build_model, default_args
    from agile3d_amd.evaluate import Evaluate
    from agile3d_amd.fit import eval_loader, fit, labelled_scenes
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = build_model(default_args()).to(dev)
    items = labelled_scenes(a.scenes, a.voxels, a.objects, colour_by_object=a.colour)
    loader, val = eval_loader(items)
    tmp = tempfile.mkdtemp(prefix="a3d_fit_")
    json.dump(val, open(os.path.join(tmp, "val.json"), "w"))
    done, opt = 0, None
    while done < a.iters:
        n = min(a.every, a.iters - done)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = fit(model, items, dev, iters=n, lr=a.lr, batch=a.batch, seed=7 + done, optimizer=opt)
        opt = fit.optimizer
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        done += n
        args = types.SimpleNamespace(output_dir=os.path.join(tmp, f"gpu{done}"), max_num_clicks=a.max_clicks,
                                     val_list=os.path.join(tmp, "val.json"))
        random.seed(11)
        import copy
        args.val_list = None
        with contextlib.redirect_stdout(io.StringIO()):
            csv = Evaluate(model, copy.deepcopy(loader), args, dev)
        rows = [l.split() for l in open(csv)]
        res = {}
        for k in (1, 2, 3, 5, 10, 15, 20):
            v = [float(r[4]) for r in rows if r[3] == f"{k}.0"]
            res[f"IoU@{k}"] = sum(v) / max(1, len(v))
        print(f"iters {done}: {1e3 * dt / n:.1f} ms/iter, loss {sum(losses[:5]) / 5:.3f} -> {sum(losses[-5:]) / 5:.3f} | "
              + " ".join(f"{k}={float(v):.3f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
