"""``torch.autograd`` glue for the training-mode forward: the reference's training loop (``engine.py:38-150``) is

    model.train();  pcd, aux, coords, pos = model.forward_backbone(data, raw_coordinates=...)
    outputs = model.forward_mask(pcd, aux, coords, pos, click_idx=..., click_time_idx=...)
    loss_dict = criterion(outputs, labels, click_weights);  losses = sum(loss_dict[k] * weight_dict[k] ...)
    optimizer.zero_grad();  losses.backward();  clip_grad_norm_(model.parameters(), 0.1);  optimizer.step()

with ``torch.optim.AdamW``.  The arithmetic of the forward AND the backward is the HIP library's (BackboneTape /
DecoderTape / a3d_mask_losses); the three ``autograd.Function``s below only tell torch which tensors the results depend
on, so that ``losses.backward()`` walks criterion -> decoder tapes -> backbone tape and leaves every parameter's
gradient in ``.grad``.  No torch kernel computes anything of consequence here.
"""
from __future__ import annotations

import torch


class _Holder:
    """Opaque argument of the Functions (tapes, click dictionaries ...): autograd passes non-tensors through."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class BackboneFn(torch.autograd.Function):
    """tape.output as a function of the backbone's parameters (``names`` gives their state-dict keys in argument order)."""

    @staticmethod
    def forward(ctx, holder, *params):
        ctx.holder = holder
        return holder.tape.output

    @staticmethod
    def backward(ctx, d_out):
        h = ctx.holder
        if getattr(h, "released", False):
            raise RuntimeError("the backbone tape was released by the first backward through it: a second backward "
                               "(retain_graph=True, or two losses back-propagated separately) is not supported -- sum the "
                               "losses and call backward once")
        grads = h.tape.backward(d_out.contiguous())
        h.tape.release()                     # torch runs a node's backward once: free the activations now (tape <-> closure cycle)
        h.released = True
        return (None,) + tuple(grads.get(n) for n in h.names)


class DecoderFn(torch.autograd.Function):
    """The per-layer logits of one batch sample as a function of its rows of pcd_features and the decoder's parameters."""

    @staticmethod
    def forward(ctx, holder, pcd_rows, *params):
        ctx.holder = holder
        return tuple(holder.tape.logits)

    @staticmethod
    def backward(ctx, *d_logits):
        h = ctx.holder
        if getattr(h, "released", False):
            raise RuntimeError("the decoder tape was released by the first backward through it: a second backward "
                               "(retain_graph=True, or two losses back-propagated separately) is not supported -- sum the "
                               "losses and call backward once")
        grads, d_pcd = h.tape.backward([None if g is None else g.contiguous() for g in d_logits])
        h.tape.release()
        h.released = True
        return (None, d_pcd) + tuple(grads.get(n) for n in h.names)


class CriterionFn(torch.autograd.Function):
    """[bce_0, dice_0, bce_1, dice_1, ...] (one pair per prediction level, averaged over the batch samples) as a function
    of every level's logits; the backward asks a3d_mask_losses for d(g_bce * bce + g_dice * dice) / d(logits)."""

    @staticmethod
    def forward(ctx, holder, *logits):
        ctx.holder = holder
        ctx.save_for_backward(*logits)
        return holder.values

    @staticmethod
    def backward(ctx, g):
        from .criterion import _losses_one
        h = ctx.holder
        logits = ctx.saved_tensors
        gl = g.detach().to(torch.float32).cpu().tolist()       # ONE read-back: the upstream gradient of every loss value
        nb = h.n_samples
        out = []
        for lvl in range(h.n_levels):
            for i in range(nb):
                _, grad = _losses_one(logits[lvl * nb + i], h.targets[i], None if h.weights is None else h.weights[i],
                                      gl[2 * lvl] / nb, gl[2 * lvl + 1] / nb, True)
                out.append(grad)
        return (None,) + tuple(out)
