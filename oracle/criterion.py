"""CPU oracle of the mask losses (models/criterion.py:14-110) in plain torch with autograd.  TEST INFRASTRUCTURE ONLY.

PINNED: tests/golden/make_criterion_goldens.py runs the reference's own SetCriterion (pure torch, loaded by path from
/root/reference) on seeded inputs; tests/test_criterion.py re-checks this file against the stored loss values and
autograd gradients (<= 1e-6)."""
import torch
import torch.nn.functional as F


def loss_bce_sample(logits, target, weight):
    """criterion.py:83-91, one sample."""
    return (F.cross_entropy(logits, target.long(), reduction="none") * weight).mean()


def loss_dice_sample(logits, target, weight, eps=1e-6):
    """criterion.py:14-81, one sample: the [N, C] matrix is treated as N items of C 'pixels' each."""
    p = logits.softmax(1)
    onehot = F.one_hot(target.long(), logits.shape[1]).to(p.dtype)
    num = 2.0 * (p * onehot).mean(1)
    den = (p + onehot).mean(1)
    soft_iou = (num + eps) / (den + eps)
    d = torch.where(num > eps, 1.0 - soft_iou, soft_iou * 0.0)
    return (d * weight).mean()


def criterion(outputs, targets, weights, losses=("bce", "dice")):
    """criterion.py:112-139: loss dict incl. the '_<i>' copies for aux_outputs."""
    fns = {"bce": loss_bce_sample, "dice": loss_dice_sample}

    def level(pred, suffix):
        out = {}
        for name in losses:
            tot = 0.0
            for i in range(len(pred)):
                tot = tot + fns[name](pred[i], targets[i], weights[i])
            out[f"loss_{name}{suffix}"] = tot / len(pred)
        return out
    d = level(outputs["pred_masks"], "")
    for i, aux in enumerate(outputs.get("aux_outputs", [])):
        d.update(level(aux["pred_masks"], f"_{i}"))
    return d


def total_and_grads(outputs, targets, weights, weight_dict, losses=("bce", "dice")):
    """engine.py:126-128: weighted total and its gradient w.r.t. every logits tensor."""
    leaves = [p.detach().clone().requires_grad_(True) for p in outputs["pred_masks"]]
    aux_leaves = [[p.detach().clone().requires_grad_(True) for p in a["pred_masks"]] for a in outputs.get("aux_outputs", [])]
    o = {"pred_masks": leaves, "aux_outputs": [{"pred_masks": a} for a in aux_leaves]}
    d = criterion(o, targets, weights, losses)
    total = sum(d[k] * weight_dict[k] for k in d if k in weight_dict)
    total.backward()
    return d, total.detach(), [l.grad for l in leaves], [[l.grad for l in a] for a in aux_leaves]
