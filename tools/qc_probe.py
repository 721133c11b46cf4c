import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from agile3d_amd import SparseTensor, build_model, default_args, lib as L, randomize_bn_stats
from agile3d_amd.synthetic import make_clicks, make_scene
torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
sc = make_scene(80000, seed=0)
ci, ct = make_clicks(sc["labels"], 5, int(os.environ.get("CPO", "2")), 0, seed=0)
coords, feats, raw = (torch.from_numpy(sc[k]).cuda() for k in ("coords", "feats", "raw_xyz"))
r = model.forward_backbone(SparseTensor(features=feats, coordinates=coords), raw_coordinates=raw)
lib = L.load()
for i in range(6):
    if i == 5:
        lib.a3d_profile_read(None, 0); lib.a3d_profile_enable(1)
    out = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
torch.cuda.synchronize(); lib.a3d_profile_enable(0)
buf = (L.ProfEntry * 256)(); n = lib.a3d_profile_read(buf, 256)
for i in range(n):
    print(L.PROF_NAMES[buf[i].id], buf[i].n_out, round(buf[i].ms * 1e3, 1), "us")
import time
torch.cuda.synchronize()
ts = []
for i in range(50):
    t0 = time.perf_counter()
    out = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
    ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host time per forward_mask call (enqueue only): median %.1f us, min %.1f us" % (1e6 * np.median(ts), 1e6 * min(ts)))
t0 = time.perf_counter()
for i in range(200):
    out = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
torch.cuda.synchronize()
print("200 back-to-back passes: %.1f us per pass" % (1e6 * (time.perf_counter() - t0) / 200))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(100):
    out = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
