"""Fit the model on a handful of seeded synthetic labelled scenes with the repository's own training path
(``train_step.train_one_step`` = the reference's ``engine.py:38-150`` iteration on the HIP kernels) -- the way
``bench.py`` and the tests obtain a state dict whose predictions are NOT random: ScanNet and the authors' checkpoint are
not available offline, and the interactive protocol on random-init weights only ever sees a few huge error clusters
(IoU ~0.05, no NoC threshold crossed).  A fitted model puts the GPU-vs-oracle comparison of ``Evaluate`` into the
regime the reference operates in: many small error clusters, distance ties, thresholds crossed mid-run.

    scenes = labelled_scenes(4, voxels=5000, objects=3)
    stats = fit(model, scenes, device, iters=300, lr=1e-3)
    loader, val = eval_loader(scenes)              # the collate format Evaluate takes + the val-list dict

Everything is seeded (numpy / torch / random) and the training iteration is deterministic, so the same call produces
the same weights on the same hardware.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from .synthetic import make_scene


def labelled_scenes(n_scenes: int, voxels: int = 5000, objects: int = 3, seed: int = 100, colour_by_object: float = 0.0):
    """``n_scenes`` synthetic scenes with the ``objects`` largest boxes as objects 1..objects and everything else as
    background 0.  ``colour_by_object`` > 0 mixes a per-box base colour into the U[0,1) colours (real scans are not
    colour-noise); 0 keeps SURVEY 8(d)'s generator untouched."""
    out = []
    for s in range(n_scenes):
        sc = make_scene(voxels, seed=seed + s)
        raw_labels = sc["labels"]
        sizes = sorted(((int((raw_labels == i).sum()), int(i)) for i in np.unique(raw_labels) if i > 0), reverse=True)
        labels = np.zeros(len(raw_labels), np.int64)
        for k, (_, i) in enumerate(sizes[:objects], start=1):
            labels[raw_labels == i] = k
        if colour_by_object > 0:
            rng = np.random.default_rng(seed + 7919 * (s + 1))
            base = rng.random((int(raw_labels.max()) + 1, 3), dtype=np.float32)
            sc["feats"] = ((1 - colour_by_object) * sc["feats"] + colour_by_object * base[raw_labels]).astype(np.float32)
        out.append({"scene": sc, "labels": labels, "objects": objects, "name": f"scene{seed + s:04d}_00"})
    return out


def train_batch(items, background_ignored: bool = True):
    """The reference's collate tuple (datasets/InterMultiObj3DSegDataset.py:126-136) for a list of scenes.
    ``background_ignored``: the background rows carry -1, the id ``engine.py:60-61`` drops from the objects an iteration
    may draw (with 0 the shell itself would be drawn as an "object" now and then)."""
    from .sparse import batched_coordinates
    coords = batched_coordinates([it["scene"]["coords"][:, 1:] for it in items])
    raw = torch.from_numpy(np.concatenate([it["scene"]["raw_xyz"] for it in items]))
    feats = torch.from_numpy(np.concatenate([it["scene"]["feats"] for it in items]))
    labels = [torch.from_numpy(np.where(it["labels"] == 0, -1, it["labels"]) if background_ignored else it["labels"])
              for it in items]
    return (coords, raw, feats, labels, None, None, [{} for _ in items], tuple(it["name"] for it in items),
            tuple(it["objects"] for it in items))


def eval_loader(items):
    """One batch per scene in the format ``Evaluate`` unpacks (labels_full = labels, identity inverse map) and the
    val-list dictionary ``EvaluatorMO`` wants."""
    loader, val = [], {}
    for it in items:
        sc, lab = it["scene"], torch.from_numpy(it["labels"])
        K = it["objects"]
        loader.append((torch.from_numpy(sc["coords"]), torch.from_numpy(sc["raw_xyz"]), torch.from_numpy(sc["feats"]),
                       [lab], [lab], [torch.arange(len(lab))], [{str(k): [] for k in range(K + 1)}], [it["name"]], [K]))
        val[f"{it['name']}_obj_{K}"] = {}
    return loader, val


def fit(model, items, device, iters: int = 300, lr: float = 1e-3, weight_decay: float = 1e-4, batch: int = 2,
        seed: int = 7, max_norm: float = 0.1, log=None, optimizer=None):
    """``iters`` iterations of ``train_one_step`` (library AdamW, clip ``max_norm``) over batches of ``batch`` scenes
    drawn round-robin.  Returns the per-iteration losses; ``optimizer`` continues an earlier call's AdamW state (it is
    left in ``fit.optimizer``)."""
    from .criterion import build_mask_criterion
    from .model import default_args
    from .optim import AdamW
    from .train_step import train_one_step
    args = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])
    criterion = build_mask_criterion(args)
    opt = optimizer if optimizer is not None else AdamW(model.named_parameters(), lr=lr, weight_decay=weight_decay)
    fit.optimizer = opt
    np.random.seed(seed), torch.manual_seed(seed), random.seed(seed)
    batches = [train_batch([items[(b * batch + j) % len(items)] for j in range(batch)])
               for b in range((len(items) + batch - 1) // batch)]
    losses = []
    for it in range(iters):
        st = train_one_step(model, criterion, opt, batches[it % len(batches)], device, max_norm=max_norm)
        losses.append(st["loss"])
        if log is not None and (it % 25 == 0 or it == iters - 1):
            log(f"fit: iteration {it} loss {st['loss']:.4f} grad_norm {st['grad_norm']:.3f} clicks {st['clicks']}")
    model.eval()
    return losses
