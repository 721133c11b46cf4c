#!/usr/bin/env python3
"""Where a lock-step evaluation round of B scenes goes (bench.py's eval_rounds_per_s protocol): wall time per part
(device-synchronised) and a host profile (cProfile) of the rounds.   python tools/eval_round_profile.py [batch] [voxels]"""
import cProfile, os, pstats, random, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agile3d_amd import SparseTensor, build_model, default_args, randomize_bn_stats, clicks as pc
from agile3d_amd.synthetic import make_scene

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
VOX = int(sys.argv[2]) if len(sys.argv) > 2 else 80_000
OBJ = 5
torch.manual_seed(0)
dev = torch.device("cuda")
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
scenes = [make_scene(VOX, seed=b, batch_index=b) for b in range(B)]
coords, feats, raw = (torch.from_numpy(np.concatenate([s[k] for s in scenes])).cuda() for k in ("coords", "feats", "raw_xyz"))
rB = model.forward_backbone(SparseTensor(features=feats, coordinates=coords), raw_coordinates=raw)
labs, raws = [], []
for s_ in scenes:
    lb = np.zeros(len(s_["coords"]), np.int64)
    sz = sorted(((int((s_["labels"] == i).sum()), i) for i in np.unique(s_["labels"]) if i > 0), reverse=True)
    for k_, (_, i) in enumerate(sz[:OBJ], start=1):
        lb[s_["labels"] == i] = k_
    labs.append(torch.from_numpy(lb).to(dev))
    raws.append(torch.from_numpy(s_["raw_xyz"]).to(dev))
ecis = [{str(k_): [] for k_ in range(OBJ + 1)} for _ in scenes]
ects = [{str(k_): [] for k_ in range(OBJ + 1)} for _ in scenes]
preds = [torch.zeros(len(s_["coords"]), dtype=torch.int32, device=dev) for s_ in scenes]
random.seed(0)
parts = {"forward_mask": [], "argmax": [], "iou + clusters": [], "pick + extend": [], "round": []}


def one_round(rnd, timed):
    global preds
    def lap(t0, key):
        if timed:
            torch.cuda.synchronize()
            parts[key].append(time.perf_counter() - t0)
        return time.perf_counter()
    torch.cuda.synchronize()
    tr = t0 = time.perf_counter()
    if rnd:
        outs = model.forward_mask(*rB, click_idx=ecis, click_time_idx=ects)["pred_masks"]
        t0 = lap(t0, "forward_mask")
        preds = pc.argmax_labels_batch(outs, ecis)
        t0 = lap(t0, "argmax")
    _, cls_ = pc.mean_iou_and_clusters_batch(preds, labs, None, labs, raws)
    t0 = lap(t0, "iou + clusters")
    for b_, (new, _, _, nt) in enumerate(pc.pick_clicks_batch(cls_, labs, raws, rnd, training=False)):
        if new is not None:
            pc.extend_clicks(ecis[b_], ects[b_], new, nt)
    lap(t0, "pick + extend")
    torch.cuda.synchronize()
    if timed:
        parts["round"].append(time.perf_counter() - tr)


for rnd in range(4):
    one_round(rnd, False)
for rnd in range(4, 12):
    one_round(rnd, True)
print(f"{B} scenes x {VOX} voxels, lock-step round, medians of 8 (each part device-synchronised):")
for k, v in parts.items():
    print(f"  {k:16s} {1e3 * float(np.median(v)):8.3f} ms")
plain = []
for rnd in range(12, 18):
    torch.cuda.synchronize(); t0 = time.perf_counter(); one_round(rnd, False); plain.append(time.perf_counter() - t0)
print(f"  unsynchronised round {1e3 * float(np.median(plain)):.3f} ms -> {B / float(np.median(plain)):.0f} scene-rounds/s")
pr = cProfile.Profile(); pr.enable()
for rnd in range(18, 22):
    one_round(rnd, False)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
