"""Mask losses (row f-2, first piece): oracle vs the reference's own SetCriterion (CPU), HIP path vs both (GPU)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import criterion as oc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "criterion_case.npz"))
WEIGHT_DICT = {"loss_bce": 1.0, "loss_dice": 2.0}
WEIGHT_DICT.update({f"{k}_{i}": v for i in range(3) for k, v in list(WEIGHT_DICT.items())[:2]})


def case(device="cpu"):
    t = [torch.from_numpy(G[f"target{i}"]).to(device) for i in range(2)]
    w = [torch.from_numpy(G[f"weight{i}"]).to(device) for i in range(2)]
    lv = [[torch.from_numpy(G[f"logits_l{l}_s{i}"]).to(device) for i in range(2)] for l in range(3)]
    return {"pred_masks": lv[0], "aux_outputs": [{"pred_masks": lv[1]}, {"pred_masks": lv[2]}]}, t, w


def test_oracle_matches_reference_goldens():
    outputs, t, w = case()
    d, total, g0, gaux = oc.total_and_grads(outputs, t, w, WEIGHT_DICT)
    for k in d:
        assert abs(float(d[k]) - float(G["loss/" + k])) <= 1e-6, k
    assert abs(float(total) - float(G["total"])) <= 1e-5
    for l, gl in enumerate([g0] + gaux):
        for i, g in enumerate(gl):
            assert np.abs(g.numpy() - G[f"grad_l{l}_s{i}"]).max() <= 1e-7


@pytest.mark.gpu
def test_hip_criterion_matches_reference_goldens():
    from agile3d_amd.criterion import build_mask_criterion
    args = types.SimpleNamespace(bce_loss_coef=1.0, dice_loss_coef=2.0, aux=True, num_decoders=3, hlevels=[4],
                                 losses=["bce", "dice"])
    crit = build_mask_criterion(args)
    assert crit.weight_dict == WEIGHT_DICT
    outputs, t, w = case("cuda")
    d = crit(outputs, t, w)
    assert set(d) == {k[5:] for k in G.files if k.startswith("loss/")}
    for k, v in d.items():
        assert abs(float(v) - float(G["loss/" + k])) <= 2e-6, (k, float(v), float(G["loss/" + k]))
    total = sum(float(d[k]) * crit.weight_dict[k] for k in d)
    assert abs(total - float(G["total"])) <= 2e-5
    g = crit.grad_logits(outputs, t, w)
    for l, gl in enumerate([g["pred_masks"]] + g["aux_outputs"]):
        for i, gi in enumerate(gl):
            ref = G[f"grad_l{l}_s{i}"]
            err = np.abs(gi.cpu().numpy() - ref).max()
            assert err <= 1e-8 + 1e-5 * np.abs(ref).max(), (l, i, err)


@pytest.mark.gpu
def test_hip_criterion_bad_target_and_subsets():
    from agile3d_amd.criterion import SetCriterion
    z = [torch.randn(100, 3, device="cuda")]
    crit = SetCriterion({"loss_bce": 1.0}, ["bce"])
    d = crit({"pred_masks": z}, [torch.zeros(100, dtype=torch.long)], [torch.ones(100)])
    assert set(d) == {"loss_bce"} and torch.isfinite(d["loss_bce"])
    bad = crit({"pred_masks": z}, [torch.full((100,), 3, dtype=torch.long)], [torch.ones(100)])
    assert torch.isnan(bad["loss_bce"])
    with pytest.raises(AssertionError):
        SetCriterion({}, ["focal"])
    with pytest.raises(RuntimeError):
        crit({"pred_masks": [torch.randn(10, 3)]}, [torch.zeros(10, dtype=torch.long)], [torch.ones(10)])
