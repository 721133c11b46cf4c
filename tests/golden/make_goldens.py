#!/usr/bin/env python3
"""Generate the committed golden vectors for the decoder path by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's ``models`` package with a stub ``MinkowskiEngine``
module (the reference's decoder code -- ``Agile3d.forward_mask``, ``mask_module``,
``get_pos_encs``, ``attention_block.py``, ``position_embedding.py`` -- is pure torch;
only module construction touches ME), loads OUR model's ``state_dict`` into the
reference model with ``strict=True`` (which pins the checkpoint key layout), feeds
duck-typed sparse tensors (``.F`` / ``.C``) through the reference's CPU branch and stores
inputs + outputs as ``.npz``.  Recipe: SURVEY.md Appendix E.

Outputs (tests/golden/):
  decoder_weights.npz         every non-backbone entry of the state dict (5.7 MB)
  decoder_case_<name>.npz     feats128, xyz, click arrays, pos_enc, logits of the 3
                              decoder iterations, the 2 intermediate attention masks
  state_dict_keys.json        key -> shape of the full reference state dict
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def install_me_stub():
    me = types.ModuleType("MinkowskiEngine")

    class MinkowskiNetwork(nn.Module):
        def __init__(self, D):
            super().__init__()
            self.D = D

    class RegionType:
        HYPER_CUBE, HYPER_CROSS, CUSTOM = 0, 1, 2

        def __init__(self, v):
            self.v = v

    class KernelGenerator:
        def __init__(self, kernel_size, stride=1, dilation=1, region_type=None, axis_types=None, dimension=3):
            ks = kernel_size if isinstance(kernel_size, (list, tuple)) else [kernel_size] * dimension
            self.kernel_volume = int(np.prod(ks))

    class _Conv(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                     kernel_generator=None, dimension=3):
            super().__init__()
            vol = kernel_generator.kernel_volume
            s = stride if isinstance(stride, int) else int(np.max(stride))
            shape = (in_channels, out_channels) if (vol == 1 and s == 1) else (vol, in_channels, out_channels)
            self.kernel = nn.Parameter(torch.zeros(*shape))
            self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None

    class MinkowskiConvolution(_Conv):
        pass

    class MinkowskiConvolutionTranspose(_Conv):
        pass

    class MinkowskiBatchNorm(nn.Module):
        def __init__(self, n, momentum=0.1, **kw):
            super().__init__()
            self.bn = nn.BatchNorm1d(n, momentum=momentum)

    class _Noop(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    for name, obj in dict(MinkowskiNetwork=MinkowskiNetwork, RegionType=RegionType,
                          KernelGenerator=KernelGenerator, MinkowskiConvolution=MinkowskiConvolution,
                          MinkowskiConvolutionTranspose=MinkowskiConvolutionTranspose,
                          MinkowskiBatchNorm=MinkowskiBatchNorm, MinkowskiReLU=_Noop,
                          MinkowskiInstanceNorm=_Noop, MinkowskiAvgPooling=_Noop,
                          MinkowskiAvgUnpooling=_Noop, MinkowskiSumPooling=_Noop, SparseTensor=_Noop).items():
        setattr(me, name, obj)
    ops = types.ModuleType("MinkowskiEngine.MinkowskiOps")
    ops.SparseTensor = _Noop
    ops.cat = lambda *a: None
    pool = types.ModuleType("MinkowskiEngine.MinkowskiPooling")
    pool.MinkowskiAvgPooling = _Noop
    me.MinkowskiOps, me.MinkowskiPooling = ops, pool
    sys.modules["MinkowskiEngine"] = me
    sys.modules["MinkowskiEngine.MinkowskiOps"] = ops
    sys.modules["MinkowskiEngine.MinkowskiPooling"] = pool


class ST:  # duck-typed sparse tensor for the reference's CPU branch (reads .F and .C only)
    def __init__(self, F, C):
        self.F, self.C = F, C


def clicks_to_arrays(click_idx, click_time):
    """Flatten the dicts: rows, object id (0 = bg), time -- in key order '0','1',..,'K'."""
    K = len(click_idx) - 1
    rows, objs, times = [], [], []
    for o in range(0, K + 1):
        for r, t in zip(click_idx[str(o)], click_time[str(o)]):
            rows.append(r)
            objs.append(o)
            times.append(t)
    return (np.array(rows, np.int32), np.array(objs, np.int32), np.array(times, np.int32), K)


def make_case(ref_model, n, K, clicks_per_obj, n_bg, seed, feat_scale=1.0, dup_click=False):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(n, 128, generator=g) * feat_scale
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([6.0, 4.0, 2.5])
    C = torch.zeros(n, 4, dtype=torch.int32)
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n)
    click_idx, click_time = {"0": []}, {"0": []}
    t, p = 0, 0
    order = []
    for o in range(1, K + 1):
        cnt = clicks_per_obj if isinstance(clicks_per_obj, int) else clicks_per_obj[o - 1]
        click_idx[str(o)] = [int(x) for x in perm[p:p + cnt]]
        p += cnt
        order += [str(o)] * cnt
    if dup_click and K >= 2:   # object 2 clicks exactly where object 1 clicked: forces an empty label
        click_idx["2"] = list(click_idx["1"][:len(click_idx["2"])])
    click_idx["0"] = [int(x) for x in perm[p:p + n_bg]]
    order += ["0"] * n_bg
    # global click times: a random interleaving, as random.shuffle in utils/seg.py:128 produces
    times = rng.permutation(len(order))
    cnt = {k: 0 for k in click_idx}
    for k in click_idx:
        click_time[k] = []
    for key, tm in zip(order, times):
        click_time[key].append(int(tm))
    coords_st = ST(xyz, C)
    recorded = []
    orig = ref_model.mask_module

    def hook(*a, **k):
        out = orig(*a, **k)
        recorded.append(out[1].clone())
        return out

    ref_model.mask_module = hook
    with torch.no_grad():
        pos = ref_model.get_pos_encs([coords_st] * 5)
        out = ref_model.forward_mask(ST(feats, C), None, coords_st, pos,
                                     click_idx=[click_idx], click_time_idx=[click_time])
    ref_model.mask_module = orig
    logits = [a["pred_masks"][0] for a in out["aux_outputs"]] + [out["pred_masks"][0]]
    rows, objs, tms, K_ = clicks_to_arrays(click_idx, click_time)
    empties = []
    for lg in logits:
        lab = lg.argmax(1)
        empties.append([int((lab == o).sum()) for o in range(K + 1)])
    return dict(feats128=feats.numpy(), xyz=xyz.numpy(), click_rows=rows, click_objs=objs,
                click_times=tms, K=np.int32(K_), pos_enc=pos[4][0][0].numpy(),
                logits0=logits[0].numpy(), logits1=logits[1].numpy(), logits2=logits[2].numpy(),
                attn_mask0=recorded[0].numpy(), attn_mask1=recorded[1].numpy(),
                label_hist=np.array(empties, np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    install_me_stub()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import models as ref_models  # noqa  (the reference)
    from agile3d_amd.model import build_model, default_args, randomize_bn_stats

    args = default_args()
    torch.manual_seed(0)
    ours = randomize_bn_stats(build_model(args)).eval()
    ref = ref_models.build_model(args).eval()
    sd = ours.state_dict()
    missing = ref.load_state_dict(sd, strict=True)   # pins the key layout AND shapes
    print("state dict loaded into the reference model:", missing)
    n_params = sum(p.numel() for p in ref.parameters())
    assert n_params == 39_289_760, n_params
    json.dump({k: list(v.shape) for k, v in ref.state_dict().items()},
              open(os.path.join(a.out, "state_dict_keys.json"), "w"), indent=0)
    dec = {k: v.numpy() for k, v in sd.items() if not k.startswith("backbone.")}
    np.savez(os.path.join(a.out, "decoder_weights.npz"), **dec)

    cases = {
        "n2048_k1": dict(n=2048, K=1, clicks_per_obj=1, n_bg=0, seed=1),
        "n2048_k3_bg": dict(n=2048, K=3, clicks_per_obj=[1, 2, 3], n_bg=2, seed=2),
        "n4096_k5x2": dict(n=4096, K=5, clicks_per_obj=2, n_bg=0, seed=3),
        "n3000_k10_bg": dict(n=3000, K=10, clicks_per_obj=[1, 2, 1, 3, 1, 1, 2, 1, 1, 4], n_bg=3, seed=4),
        "n1024_k2_dup": dict(n=1024, K=2, clicks_per_obj=1, n_bg=0, seed=5, dup_click=True),
        "n777_k4_ragged": dict(n=777, K=4, clicks_per_obj=[5, 1, 1, 2], n_bg=1, seed=6, feat_scale=3.0),
        "n48_k10_tiny": dict(n=48, K=10, clicks_per_obj=1, n_bg=0, seed=7),
        "n100_k10_tiny": dict(n=100, K=10, clicks_per_obj=2, n_bg=4, seed=8, feat_scale=0.3),
        # more than 64 queries (clicks + 10 learned background queries): the multi-object protocol adds clicks up to
        # num_obj x 20 (eval_multi_obj.py:70,116-118).  One case per tile count of the fused wide tier (decoder_wide.h):
        # 46 / 62 queries (the 33 .. 64 range: both tiers serve it) and 75 .. 205; q80, q144 and q205 hit the all-True-row rule (an object without points after iteration 0 or 1)
        "n2000_k6_q46": dict(n=2000, K=6, clicks_per_obj=5, n_bg=6, seed=16),
        "n1800_k8_q62": dict(n=1800, K=8, clicks_per_obj=6, n_bg=4, seed=17, feat_scale=0.7),
        "n1500_k7_q75": dict(n=1500, K=7, clicks_per_obj=9, n_bg=2, seed=9),
        "n100_k10_q90": dict(n=100, K=10, clicks_per_obj=8, n_bg=0, seed=10, feat_scale=0.3),
        "n800_k5_q105": dict(n=800, K=5, clicks_per_obj=18, n_bg=5, seed=11),
        "n1100_k9_q124": dict(n=1100, K=9, clicks_per_obj=12, n_bg=6, seed=12, feat_scale=0.5),
        "n1200_k8_q138": dict(n=1200, K=8, clicks_per_obj=15, n_bg=8, seed=13, dup_click=True),
        "n1000_k9_q180": dict(n=1000, K=9, clicks_per_obj=18, n_bg=8, seed=14),
        "n220_k10_q205": dict(n=220, K=10, clicks_per_obj=19, n_bg=5, seed=27, feat_scale=0.3),
        "n80_k10_q80": dict(n=80, K=10, clicks_per_obj=7, n_bg=0, seed=20, feat_scale=0.3),
        "n150_k12_q144": dict(n=150, K=12, clicks_per_obj=11, n_bg=2, seed=20, feat_scale=0.3),
    }
    for name, kw in cases.items():
        c = make_case(ref, **kw)
        np.savez(os.path.join(a.out, f"decoder_case_{name}.npz"), **c)
        print(name, "label histogram per iteration:", c["label_hist"].tolist(),
              "| empty-label rule hit:", bool((c["label_hist"][:2] == 0).any()))


if __name__ == "__main__":
    main()
