#!/usr/bin/env python3
"""Time of one click-simulator round (a3d_click_clusters: error clusters + nearest-outside-point search) on a synthetic
scene with a given fraction of wrongly labelled points:  python tools/click_bench.py [voxels] [wrong_fraction]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agile3d_amd import clicks as pc
from agile3d_amd.synthetic import make_scene

n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 80_000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
sc = make_scene(n_vox, seed=0)
lab_np = np.zeros(len(sc["labels"]), np.int64)
sizes = sorted(((int((sc["labels"] == i).sum()), i) for i in np.unique(sc["labels"]) if i > 0), reverse=True)
for k, (_, i) in enumerate(sizes[:5], start=1):
    lab_np[sc["labels"] == i] = k
lab = torch.from_numpy(lab_np).cuda()
rng = np.random.default_rng(0)
pred_np = lab_np.copy()
# wrong points in spatially coherent blobs: whole slabs of the scene get the label of another object
xyz = sc["raw_xyz"]
order = np.argsort(xyz[:, 0] + 0.3 * xyz[:, 1])
nwrong = int(frac * len(order))
start = len(order) // 5
sel = order[start:start + nwrong]
pred_np[sel] = (lab_np[sel] + 1 + (np.arange(len(sel)) // max(1, len(sel) // 3))) % 6
pred = torch.from_numpy(pred_np).int().cuda()
raw = torch.from_numpy(xyz).float().cuda()
print("voxels", len(lab_np), "wrong", int((pred_np != lab_np).sum()))
for _ in range(3):
    pc.get_simulated_clicks(pred, lab, raw, 1, training=False)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    pc.get_simulated_clicks(pred, lab, raw, 1, training=False)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print("get_simulated_clicks: median %.3f ms (min %.3f)" % (float(np.median(ts)), min(ts)))
