"""Host mirror of the reference's ``models/criterion.py`` on the HIP library:
same class name, constructor, ``forward(outputs, targets, weights)`` and loss-dict keys (``loss_bce``, ``loss_dice``
and their ``_<i>`` copies for ``aux_outputs``).  Besides the loss values there is ``grad_logits`` -- the gradient of
the weighted total (engine.py:126-128) with respect to every prediction level's logits, which is where
``train_step.train_one_step`` starts the backward pass.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def _losses_one(logits, target, weight, coef_bce, coef_dice, want_grad):
    lib = L.load()
    if not logits.is_cuda:
        raise RuntimeError("agile3d_amd.criterion runs on the GPU only (no CPU fallback)")
    z = logits.to(torch.float32).contiguous()
    t = target.to(device=z.device, dtype=torch.int32).contiguous()
    w = None if weight is None else weight.to(device=z.device, dtype=torch.float32).contiguous()
    n, c = z.shape
    out = torch.empty(2, dtype=torch.float32, device=z.device)
    grad = torch.empty_like(z) if want_grad else None
    ws = torch.empty(64, dtype=torch.uint8, device=z.device)
    L.check(lib.a3d_mask_losses(z.data_ptr(), t.data_ptr(), w.data_ptr() if w is not None else None, n, c,
                                float(coef_bce), float(coef_dice), out.data_ptr(),
                                grad.data_ptr() if grad is not None else None, ws.data_ptr(), ws.numel(),
                                C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)), "a3d_mask_losses")
    return out, grad


class SetCriterion:
    """models/criterion.py:7-140.  ``losses`` is a subset of ('bce', 'dice')."""

    def __init__(self, weight_dict, losses):
        for l in losses:
            if l not in ("bce", "dice"):
                raise AssertionError(f"do you really want to compute {l} loss?")
        self.weight_dict = weight_dict
        self.losses = list(losses)
        self.training = True

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def _level(self, pred_masks, targets, weights, suffix, want_grad):
        wb = self.weight_dict.get("loss_bce" + suffix, 0.0) if "bce" in self.losses else 0.0
        wd = self.weight_dict.get("loss_dice" + suffix, 0.0) if "dice" in self.losses else 0.0
        nb = len(pred_masks)
        outs, grads = [], []
        for i in range(nb):
            out, g = _losses_one(pred_masks[i], targets[i], None if weights is None else weights[i], wb / nb, wd / nb,
                                 want_grad)
            outs.append(out)
            grads.append(g)
        tot = outs[0]
        for o in outs[1:]:                # sample order, like the reference's running sum (models/criterion.py:93-101)
            tot = tot + o
        tot = tot / nb
        d = {}
        if "bce" in self.losses:
            d["loss_bce" + suffix] = tot[0]
        if "dice" in self.losses:
            d["loss_dice" + suffix] = tot[1]
        return d, grads

    def __call__(self, outputs, targets, weights=None):
        return self.forward(outputs, targets, weights)

    def forward(self, outputs, targets, weights=None):
        levels = [("", outputs["pred_masks"])] + [(f"_{i}", aux["pred_masks"]) for i, aux in enumerate(outputs.get("aux_outputs", []))]
        tracked = torch.is_grad_enabled() and any(p.requires_grad for _, preds in levels for p in preds)
        losses, vals = {}, []
        for suffix, preds in levels:
            d, _ = self._level([p.detach() for p in preds], targets, weights, suffix, False)
            losses.update(d)
            vals += [d.get("loss_bce" + suffix), d.get("loss_dice" + suffix)]
        if not tracked:
            return losses
        # training through torch.autograd (engine.py:128-147: losses.backward()): the loss values become the outputs of
        # ONE autograd node over every level's logits; its backward is a3d_mask_losses again (autograd.CriterionFn)
        from .autograd import CriterionFn, _Holder
        zero = torch.zeros((), dtype=torch.float32, device=levels[0][1][0].device)
        h = _Holder(values=torch.stack([zero if v is None else v for v in vals]), targets=list(targets),
                    weights=None if weights is None else list(weights), n_levels=len(levels), n_samples=len(levels[0][1]))
        flat = [p for _, preds in levels for p in preds]
        out = CriterionFn.apply(h, *flat)
        for l, (suffix, _) in enumerate(levels):
            if "loss_bce" + suffix in losses:
                losses["loss_bce" + suffix] = out[2 * l]
            if "loss_dice" + suffix in losses:
                losses["loss_dice" + suffix] = out[2 * l + 1]
        return losses

    def forward_and_grad(self, outputs, targets, weights=None):
        """``forward`` and ``grad_logits`` from ONE pass of a3d_mask_losses per level and sample (the kernel produces the
        loss values and the gradient together): -> (loss dict of device scalars, the structure ``grad_logits`` returns)."""
        levels = [("", outputs["pred_masks"])] + [(f"_{i}", aux["pred_masks"]) for i, aux in enumerate(outputs.get("aux_outputs", []))]
        losses, res = {}, {"pred_masks": None, "aux_outputs": []}
        for li, (suffix, preds) in enumerate(levels):
            d, g = self._level([p.detach() for p in preds], targets, weights, suffix, True)
            losses.update(d)
            if li == 0:
                res["pred_masks"] = g
            else:
                res["aux_outputs"].append(g)
        return losses, res

    def grad_logits(self, outputs, targets, weights=None):
        """{'pred_masks': [grad per sample], 'aux_outputs': [[grad per sample] per level]}: gradient of
        sum_k weight_dict[k] * loss_dict[k] (engine.py:128) w.r.t. the logits of every level."""
        _, g = self._level(outputs["pred_masks"], targets, weights, "", True)
        res = {"pred_masks": g, "aux_outputs": []}
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            _, ga = self._level(aux["pred_masks"], targets, weights, f"_{i}", True)
            res["aux_outputs"].append(ga)
        return res


def build_mask_criterion(args):
    """models/criterion.py:142-155."""
    weight_dict = {"loss_bce": args.bce_loss_coef, "loss_dice": args.dice_loss_coef}
    if args.aux:
        aux = {}
        for i in range(args.num_decoders * len(args.hlevels)):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    return SetCriterion(weight_dict, args.losses)
