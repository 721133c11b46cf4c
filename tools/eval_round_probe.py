"""Where a lock-step evaluation round goes (eval_multi_obj.py:112-166 with a batch): the parts of bench.py's
`eval_rounds_per_s` loop timed one by one (device-synchronised) next to the round as it runs, plus the host time of the
enqueue alone.  ER_BATCH scenes (16), ER_ROUNDS rounds (16)."""
import os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from agile3d_amd import SparseTensor, build_model, default_args, randomize_bn_stats
from agile3d_amd import clicks as pc
from agile3d_amd.synthetic import make_scene

B = int(os.environ.get("ER_BATCH", "16"))
R = int(os.environ.get("ER_ROUNDS", "16"))
K = 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().to(dev)
scenes = [make_scene(80000, seed=b, batch_index=b) for b in range(B)]
coords = torch.from_numpy(np.concatenate([s_["coords"] for s_ in scenes])).to(dev)
feats = torch.cat([torch.from_numpy(s["feats"]) for s in scenes]).to(dev)
raw = torch.cat([torch.from_numpy(s["raw_xyz"]) for s in scenes]).to(dev)
rB = model.forward_backbone(SparseTensor(features=feats, coordinates=coords), raw_coordinates=raw)
labs, raws = [], []
for s_ in scenes:
    lb = np.zeros(len(s_["coords"]), np.int64)
    sz = sorted(((int((s_["labels"] == i).sum()), i) for i in np.unique(s_["labels"]) if i > 0), reverse=True)
    for k_, (_, i) in enumerate(sz[:K], start=1):
        lb[s_["labels"] == i] = k_
    labs.append(torch.from_numpy(lb).to(dev))
    raws.append(torch.from_numpy(s_["raw_xyz"]).to(dev))


def run(sync_parts):
    ecis = [{str(k_): [] for k_ in range(K + 1)} for _ in scenes]
    ects = [{str(k_): [] for k_ in range(K + 1)} for _ in scenes]
    preds = [torch.zeros(len(s_["coords"]), dtype=torch.int32, device=dev) for s_ in scenes]
    random.seed(0)
    parts = {"forward_mask": [], "argmax": [], "iou+clusters": [], "pick+extend": [], "round": []}
    for rnd in range(R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t = t0
        def lap(name):
            nonlocal t
            if sync_parts:
                torch.cuda.synchronize()
            now = time.perf_counter()
            if rnd >= 3:
                parts[name].append(1e3 * (now - t))
            t = now
        if rnd:
            outs = model.forward_mask(*rB, click_idx=ecis, click_time_idx=ects)["pred_masks"]
        lap("forward_mask")
        if rnd:
            preds = pc.argmax_labels_batch(outs, ecis)
        lap("argmax")
        _, cls_ = pc.mean_iou_and_clusters_batch(preds, labs, None, labs, raws)
        lap("iou+clusters")
        for b_, (new, _, _, nt) in enumerate(pc.pick_clicks_batch(cls_, labs, raws, rnd, training=False)):
            if new is not None:
                pc.extend_clicks(ecis[b_], ects[b_], new, nt)
        lap("pick+extend")
        torch.cuda.synchronize()
        if rnd >= 3:
            parts["round"].append(1e3 * (time.perf_counter() - t0))
    return {k: float(np.median(v)) for k, v in parts.items()}


for sync_parts in (True, False, True, False):
    p = run(sync_parts)
    print(("parts synchronised:   " if sync_parts else "as it runs (host ms): ") +
          "  ".join(f"{k} {v:.3f}" for k, v in p.items()) + f"   -> {B / p['round'] * 1e3:.0f} scene-rounds/s")
