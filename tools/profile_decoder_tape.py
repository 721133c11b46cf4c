#!/usr/bin/env python3
"""Host-side profile (cProfile) of one DecoderTape forward + backward on an 80 k-voxel scene: where the Python time of the
training decoder goes.  python tools/profile_decoder_tape.py"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agile3d_amd import build_model, default_args
from agile3d_amd.synthetic import make_clicks, make_scene
from agile3d_amd.train_decoder import DecoderTape

torch.manual_seed(0)
model = build_model(default_args()).cuda().train()
sc = make_scene(80_000, seed=0)
n = len(sc["coords"])
ci, ct = make_clicks(sc["labels"], 5, 3, 2, seed=0)
pcd = torch.randn(n, 128, device="cuda") * 0.3
pos = torch.randn(n, 128, device="cuda") * 0.3
K = len(ci) - 1
R = [torch.randn(n, K + 1, device="cuda") / 8 for _ in range(3)]
def once():
    tp = DecoderTape(model, pcd, pos, ci, ct)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tp.backward(R)
    torch.cuda.synchronize()
    return t1
for _ in range(2): once()
torch.cuda.synchronize(); t0 = time.perf_counter(); t1 = once(); t2 = time.perf_counter()
print(f"forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms (one sample, wall incl. device sync)")
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
