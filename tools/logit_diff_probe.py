#!/usr/bin/env python3
"""Where a GPU-vs-oracle logit difference on a FITTED state dict comes from (backbone features or decoder):
   python tools/logit_diff_probe.py [fit_iters=60] [scene=0] [round=3]
Fits like tools/fork_hunt.py, runs the GPU protocol to the given round of the scene, then compares, on that round's clicks,
the GPU logits with the oracle's -- whole path, and the GPU decoder on the ORACLE's backbone features."""
import contextlib, io, json, os, random, sys, tempfile, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agile3d_amd import SparseTensor, build_model, default_args
from agile3d_amd.evaluate import Evaluate
from agile3d_amd.fit import eval_loader, fit, labelled_scenes
from oracle import backbone as ob, decoder as od

fit_iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scene = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(default_args()).to(dev)
items = labelled_scenes(4, 5000, 3)
fit(model, items, dev, iters=fit_iters, lr=1e-3, batch=2, seed=7)
model.eval()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
loader, val = eval_loader(items)
tmp = tempfile.mkdtemp(prefix="a3d_probe_")
json.dump(val, open(os.path.join(tmp, "val.json"), "w"))
args = types.SimpleNamespace(output_dir=os.path.join(tmp, "gpu"), max_num_clicks=20, val_list=os.path.join(tmp, "val.json"))
log = []
random.seed(11)
with contextlib.redirect_stdout(io.StringIO()):
    Evaluate(model, loader, args, dev, lambda idx, cur, pred, iou, ci_, ct_: log.append(
        (cur, float(iou), {k: list(v) for k, v in ci_.items()}, pred.cpu().clone().long(), {k: list(v) for k, v in ct_.items()})))
per = len(log) // len(items)
a = log[scene * per + rnd]
ci, ct = a[2], a[4]
print("clicks", ci)
sc = items[scene]["scene"]
xyz = torch.from_numpy(sc["raw_xyz"])
rb = ob.forward_backbone(sd, sc["coords"], torch.from_numpy(sc["feats"]), xyz)
x = SparseTensor(features=torch.from_numpy(sc["feats"]), coordinates=torch.from_numpy(sc["coords"]), device=dev)
bo = model.forward_backbone(x, raw_coordinates=xyz.to(dev))
fd = (bo[0].F.cpu() - rb["pcd_features"]).abs()
print(f"backbone features: max|diff| {fd.max():.3e} scale {rb['pcd_features'].abs().max():.2f}")
out = model.forward_mask(*bo, click_idx=[ci], click_time_idx=[ct])
got = [o["pred_masks"][0].cpu() for o in out["aux_outputs"]] + [out["pred_masks"][0].cpu()]
ref = od.forward_mask(sd, rb["pcd_features"], xyz, rb["pos_enc"], ci, ct)
for l in range(3):
    d = (got[l] - ref[l]).abs()
    print(f"whole path, layer {l}: max|diff| {d.max():.3e} scale {ref[l].abs().max():.2f} labels differ at {(got[l].argmax(-1) != ref[l].argmax(-1)).sum().item()} points")
eng = model._get_engine()
di = eng.decoder_inputs(rb["pcd_features"], xyz)
out2 = model.forward_mask(*di, click_idx=[ci], click_time_idx=[ct])
got2 = [o["pred_masks"][0].cpu() for o in out2["aux_outputs"]] + [out2["pred_masks"][0].cpu()]
for l in range(3):
    d = (got2[l] - ref[l]).abs()
    print(f"GPU decoder on the oracle's features, layer {l}: max|diff| {d.max():.3e}")
# float64 oracle decoder on the same inputs: how far is each fp32 side from it?
sd64 = {k: v.double() for k, v in sd.items()}
try:
    ref64 = od.forward_mask(sd64, rb["pcd_features"].double(), xyz.double(), rb["pos_enc"].double(), ci, ct)
    for l in range(3):
        print(f"layer {l}: |oracle32 - oracle64| {(ref[l].double() - ref64[l]).abs().max():.3e}   |GPU(oracle feats) - oracle64| {(got2[l].double() - ref64[l]).abs().max():.3e}")
except Exception as e:
    print("float64 oracle failed:", repr(e)[:200])
