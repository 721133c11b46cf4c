#!/usr/bin/env python3
"""Does replaying the backbone program of ONE scene from a captured graph beat launching its ~60 kernels eagerly?
(a probe: a graph is tied to one scene's tables and sizes, so it is not a product path)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agile3d_amd import build_model, default_args, randomize_bn_stats
from agile3d_amd.engine import Scene
from agile3d_amd.synthetic import make_scene

torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
eng = model._get_engine()
eng.refresh_weights_if_stale()
sc = make_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 80000, seed=0)
coords, feats = torch.from_numpy(sc["coords"]).cuda(), torch.from_numpy(sc["feats"]).cuda()
scene = Scene(coords)
out = torch.empty((len(coords), 128), device="cuda")
def timeit(f, n=50):
    for _ in range(5): f()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
keep = []
def eager():
    keep[:] = [eng.program.run(scene, feats, out)]
print(f"eager program: {timeit(eager):.3f} ms")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eager(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ws = eng.program.run(scene, feats, out)
torch.cuda.synchronize()
print(f"graph replay : {timeit(g.replay):.3f} ms")
