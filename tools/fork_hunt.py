#!/usr/bin/env python3
"""Look for fitted state dicts on which the interactive protocols of the GPU product and of the CPU oracle pick different
clicks, and print what bench.oracle_protocol_synced makes of every difference -- float64 distances / cluster sizes next to
both arithmetics: the evidence behind "a fork is a tie, not a bug".  Every configuration is one bench.iou_at_k call (fit on
the GPU, the GPU protocol, the oracle next to its log: EVERY round compared, the oracle continues from the GPU's clicks).
   python tools/fork_hunt.py [fit_iters ...]            (default: 40 60 80 100 140)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    iters = [int(a) for a in sys.argv[1:]] or [40, 60, 80, 100, 140]
    dev = torch.device("cuda")
    for fi in iters:
        r = bench.iou_at_k(dev, n_scenes=4, voxels=5000, objects=3, max_clicks=20, fit_iters=fi, lr=1e-3, min_iou5=None)
        f = r["forks"]
        events = [dict(scene=s["scene"], **e) for s in f["scenes"] for e in s["events"]]
        print(json.dumps({"fit_iters": fi, "iou_gpu": r["gpu"], "iou_oracle": r["oracle"], "rounds": r["rounds"],
                          "identical_clicks": r["rounds_with_identical_clicks"], "identical_iou": r["rounds_with_identical_iou"],
                          "compared_rounds": f["compared_rounds"], "identical_compared": f["identical_rounds"],
                          "click_forks": f["click_forks"], "unexplained": f["unexplained"], "events": events}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
