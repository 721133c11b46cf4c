#!/usr/bin/env python3
"""Per-launch table of one bench step: kind, level, Cin->Cout, rows, split-K, ms, algorithmic TFLOP/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from agile3d_amd import SparseTensor, build_model, default_args, lib as L, randomize_bn_stats
from agile3d_amd.engine import Scene
from agile3d_amd.synthetic import make_clicks, make_scene

torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
import numpy as np
B = int(os.environ.get("LT_BATCH", "1"))
VOX = int(os.environ.get("LT_VOXELS", "80000"))      # LT_VOXELS=300000 LT_CPO=4: BASELINE.json config 5 (300 k voxels, 20 clicks)
CPO = int(os.environ.get("LT_CPO", "2"))
scs = [make_scene(VOX, seed=b, batch_index=b) for b in range(B)]
cl = [make_clicks(s["labels"], 5, CPO, 0, seed=b) for b, s in enumerate(scs)]
cis, cts = [c[0] for c in cl], [c[1] for c in cl]
coords, feats, raw = (torch.from_numpy(np.concatenate([s[k] for s in scs])).cuda() for k in ("coords", "feats", "raw_xyz"))
def step():
    r = model.forward_backbone(SparseTensor(features=feats, coordinates=coords), raw_coordinates=raw)
    return model.forward_mask(*r, click_idx=cis, click_time_idx=cts)
for _ in range(3): step()
scn = Scene(coords)
pairs = {"n": scn.n, "conv3": []}
for lvl in range(5):
    npad = (max(scn.n[lvl], 1) + 127) // 128 * 128
    nb = scn.table(lvl, L.TAB_NBR27).reshape(27, npad)
    pairs["conv3"].append(int((nb[:, :scn.n[lvl]] < scn.n[lvl]).sum()))
print("level sizes", scn.n, "3^3 pairs", pairs["conv3"])
lib = L.load()
reps = 5
lib.a3d_profile_read(None, 0); lib.a3d_profile_enable(1)
for _ in range(reps): step()
torch.cuda.synchronize(); lib.a3d_profile_enable(0)
buf = (L.ProfEntry * 20000)(); n = lib.a3d_profile_read(buf, 20000)
per = n // reps
kinds = {L.OP_STEM: "stem", L.OP_CONV3: "conv3", L.OP_DOWN: "down", L.OP_UP: "up", L.OP_LINEAR: "lin"}
tot = 0.0
for i in range(per):
    ms = sum(buf[i + r * per].ms for r in range(reps)) / reps
    e = buf[i]; tot += ms
    name = L.PROF_NAMES[e.id]
    if e.id == 0:
        fl = bench.algorithmic_flops(e, pairs)
        print(f"{i:3d} {'deep  ' if e.bn >= 1000 else 'spconv'}<{e.bn % 1000:3d}> {kinds.get(e.table,'?'):5s} L{e.level} {e.cin:4d}->{e.cout:4d} K={e.kernel_volume & 255:3d}{'+p' if (e.kernel_volume >> 8) & 0xfff else '+h' if e.kernel_volume >> 20 else '  '} rows={e.n_out:6d} "
              f"ch={e.ksplit:3d} {1e3*ms:8.1f} us {fl/ms/1e9:7.1f} TF/s")
    elif e.id == L.PROF_DENSE:
        fl = 2.0 * e.n_out * e.cin * e.cout
        by = 4.0 * e.n_out * (e.cin + e.cout)
        print(f"{i:3d} k_dense     lin   L{e.level} {e.cin:4d}->{e.cout:4d} rows={e.n_out:6d} {1e3*ms:8.1f} us {fl/ms/1e9:7.1f} TF/s "
              f"{by/ms/1e6:7.0f} GB/s (X+Y only)")
    else:
        print(f"{i:3d} {name:18s} rows={e.n_out:6d} {1e3*ms:8.1f} us")
print("sum", tot, "ms")
