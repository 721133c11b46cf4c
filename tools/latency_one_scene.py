#!/usr/bin/env python3
"""The batch-1 / one-stream protocol of SURVEY 8(d) on its own (bench.py's latency_ms_per_scene): scene build + forward_backbone
+ one forward_mask for ONE 80 k-voxel scene at a time, host wall with a device sync on both sides.  For rocprofv3 --kernel-trace
(tools/rocprof_timeline.py: true kernel durations and the gaps between them) and for A/B runs of the small-level kernel.
  python tools/latency_one_scene.py [--voxels N] [--iters K] [--deep 0|1]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agile3d_amd import SparseTensor, build_model, default_args, lib as L, randomize_bn_stats
from agile3d_amd.synthetic import make_clicks, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--voxels", type=int, default=80000)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--deep", type=int, default=-1)
a = ap.parse_args()
torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
lib = L.load()
if a.deep >= 0:
    lib.a3d_conv_deep_mode(a.deep)
sc = make_scene(a.voxels, seed=0)
ci, ct = make_clicks(sc["labels"], 5, 2, 0, seed=0)
coords, feats, raw = (torch.from_numpy(sc[k]).cuda() for k in ("coords", "feats", "raw_xyz"))
def step():
    r = model.forward_backbone(SparseTensor(features=feats, coordinates=coords), raw_coordinates=raw)
    return model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
for _ in range(10):
    step()
ts = []
for _ in range(a.iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print(f"one scene ({len(coords)} voxels), deep mode {lib.a3d_conv_deep_mode(-1)}: median {np.median(ts):.3f} ms, min {ts.min():.3f}, p90 {np.percentile(ts, 90):.3f}")
