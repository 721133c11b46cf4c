#!/usr/bin/env python3
"""Time the interactive loop (forward_mask + argmax + IoU + click simulator) per round on one scene."""
import argparse, os, random, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agile3d_amd import SparseTensor, build_model, default_args, randomize_bn_stats, clicks as pc
from agile3d_amd.synthetic import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--voxels", type=int, default=80000)
ap.add_argument("--objects", type=int, default=5)
ap.add_argument("--rounds", type=int, default=40)
a = ap.parse_args()
torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
sc = make_scene(a.voxels, seed=0)
n = len(sc["coords"])
sizes = sorted(((int((sc["labels"] == i).sum()), i) for i in np.unique(sc["labels"]) if i > 0), reverse=True)
labels = np.zeros(n, np.int64)
for k, (_, i) in enumerate(sizes[:a.objects], start=1):
    labels[sc["labels"] == i] = k
lab = torch.from_numpy(labels).cuda()
raw = torch.from_numpy(sc["raw_xyz"]).cuda()
x = SparseTensor(features=torch.from_numpy(sc["feats"]), coordinates=torch.from_numpy(sc["coords"]), device="cuda")
bb = model.forward_backbone(x, raw_coordinates=raw)
ci = {str(k): [] for k in range(a.objects + 1)}
ct = {str(k): [] for k in range(a.objects + 1)}
pred = torch.zeros(n, dtype=torch.int32, device="cuda")
random.seed(0)
T = {"mask": 0.0, "argmax": 0.0, "iou": 0.0, "clicks": 0.0}
def tick():
    torch.cuda.synchronize(); return time.perf_counter()
for r in range(a.rounds):
    t0 = tick()
    if r:
        out = model.forward_mask(*bb, click_idx=[ci], click_time_idx=[ct])["pred_masks"][0]
        t1 = tick(); T["mask"] += t1 - t0
        pred = pc.argmax_labels(out, ci)
        t2 = tick(); T["argmax"] += t2 - t1
    else:
        t2 = t0
    iou, _ = pc.mean_iou_scene(pred, lab)
    t3 = tick(); T["iou"] += t3 - t2
    new, _, _, nt = pc.get_simulated_clicks(pred, lab, raw, r, training=False)
    t4 = tick(); T["clicks"] += t4 - t3
    if new is not None:
        pc.extend_clicks(ci, ct, new, nt)
nq = sum(len(v) for v in ci.values())
print({k: round(1e3 * v / a.rounds, 3) for k, v in T.items()}, "ms/round;", nq, "clicks at the end; IoU", float(iou))

# GPU time of the click-simulator kernels alone (HIP events around a3d_click_clusters)
from agile3d_amd import lib as L
lib = L.load()
lib.a3d_profile_read(None, 0); lib.a3d_profile_enable(1)
pred0 = torch.zeros(n, dtype=torch.int32, device="cuda")
for p_ in (pred0, pred):
    for _ in range(5):
        pc.error_clusters(p_, lab, raw)
torch.cuda.synchronize(); lib.a3d_profile_enable(0)
buf = (L.ProfEntry * 64)(); k = lib.a3d_profile_read(buf, 64)
ms = [buf[i].ms for i in range(k) if buf[i].id == 10]
print("click kernels: round-0 prediction %.3f ms, last prediction %.3f ms (n_err %d / %d)" % (
    sum(ms[:5]) / 5, sum(ms[5:]) / 5, int((pred != lab).sum()), n))
