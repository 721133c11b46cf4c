"""CPU oracle of the interactive loop around forward_mask: IoU, click simulator, loss weights.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates ``utils/seg.py`` of the reference in
plain torch-CPU; PINNED: ``tests/golden/make_click_goldens.py`` runs the reference's own
``utils/seg.py`` (pure torch, imported from /root/reference in the build container) on seeded
inputs and ``tests/test_oracle_clicks.py`` re-checks this file against the stored outputs
(click rows, click order under the same ``random.seed``, IoU bits, weights).

Reference lines followed: mean_iou_scene utils/seg.py:44-59, loss_weights :62-70,
cal_click_loss_weights :72-89, get_next_click_coo_torch :93-117,
get_next_simulated_click_multi :119-154, measure_error_size :157-171,
get_simulated_clicks :173-226, extend_clicks :229-239.
"""
from __future__ import annotations

import random

import numpy as np
import torch


def iou_single(pred_is_obj: torch.Tensor, label_is_obj: torch.Tensor) -> torch.Tensor:
    """utils/seg.py:10-18 -- intersection / union of two boolean masks (int64 / int64 -> fp32)."""
    inter = (pred_is_obj & label_is_obj).sum()
    union = pred_is_obj.sum() + label_is_obj.sum() - inter
    return inter / union


def mean_iou_scene(pred: torch.Tensor, labels: torch.Tensor):
    """utils/seg.py:44-59 -- mean over the non-zero object ids present in ``labels``."""
    ids = torch.unique(labels)
    ids = ids[ids != 0]
    total = 0.0
    per_obj = {}
    for oid in ids:
        v = iou_single(pred == oid, labels == oid)
        per_obj[int(oid)] = float(v)
        total = total + v
    total = total / len(ids)
    return total, per_obj


def loss_weights(points: torch.Tensor, clicks: torch.Tensor, tita: float, alpha: float, beta: float):
    """utils/seg.py:62-70."""
    d = torch.cdist(points, clicks).min(dim=1).values
    return alpha + (beta - alpha) * (1 - torch.clamp(d, max=tita) / tita)


def click_loss_weights(raw_coords: torch.Tensor, click_idx: dict, alpha=0.8, beta=2.0, tita=0.3):
    """One sample of cal_click_loss_weights (utils/seg.py:72-89): clicks of all objects, dict order."""
    rows = [int(r) for v in click_idx.values() for r in v]
    return loss_weights(raw_coords, raw_coords[rows], tita, alpha, beta)


def outside_distance(coords: torch.Tensor, in_cluster: torch.Tensor):
    """measure_error_size (utils/seg.py:157-171): for every cluster point the distance to the nearest
    point outside the cluster, in cluster-local order; None if either side is empty."""
    if int(in_cluster.sum()) == 0 or int((~in_cluster).sum()) == 0:
        return None
    return torch.cdist(coords[~in_cluster], coords[in_cluster]).min(dim=0).values


def error_clusters(pred: torch.Tensor, labels: torch.Tensor, coords: torch.Tensor):
    """The per-cluster part of get_simulated_clicks (utils/seg.py:186-211).

    Returns a list (ascending cluster id, = torch.unique order) of dicts
    {cluster_id, row, label, pred, error_size}: ``row`` is the first cluster point (global row)
    attaining the largest outside distance -- what get_next_click_coo_torch (:93-118) picks."""
    lab = labels.float()
    prd = pred.float()
    wrong = (prd - lab).abs() > 0
    if int(wrong.sum()) == 0:
        return []
    cid_all = lab * 96 + prd * 11
    marks = torch.full((coords.shape[0],), -1.0)
    marks[wrong] = cid_all[wrong]
    out = []
    for cid in torch.unique(cid_all[wrong]):
        member = marks == cid
        d = outside_distance(coords, member)
        if d is None:
            raise RuntimeError("error cluster covers the whole sample (the reference fails here too)")
        local = int(torch.where(d == d.max())[0][0])
        row = int(torch.nonzero(member)[local][0])
        out.append({"cluster_id": int(cid), "row": row, "label": int(lab[row]), "pred": int(prd[row]),
                    "error_size": d.max().tolist()})
    return out


def get_simulated_clicks(pred, labels, coords, current_num_clicks=None, training=True):
    """utils/seg.py:173-226 (+ :119-154).  Uses the global ``random`` stream exactly once
    (``random.shuffle`` of the selected cluster ids), as the reference does."""
    clusters = error_clusters(pred, labels, coords)
    if not clusters:
        return None, None, None, None
    sizes = {c["cluster_id"]: c["error_size"] for c in clusters}
    by_id = {c["cluster_id"]: c for c in clusters}
    ranked = sorted(sizes, key=sizes.get, reverse=True)
    if training:
        num_obj = int((torch.unique(labels.float()) != 0).sum())
        chosen = ranked[:num_obj] if len(ranked) >= num_obj else ranked
    else:
        chosen = ranked if current_num_clicks == 0 else ranked[:1]
    random.shuffle(chosen)
    new_clicks, new_pos, new_time = {}, {}, {}
    for order, cid in enumerate(chosen):
        c = by_id[cid]
        key = str(c["label"])
        new_clicks.setdefault(key, []).append(c["row"])
        new_pos.setdefault(key, []).append(coords[c["row"]])
        new_time.setdefault(key, []).append(order)
    return new_clicks, len(chosen), new_pos, new_time


def extend_clicks(current_clicks, current_clicks_time, new_clicks, new_click_time):
    """utils/seg.py:229-239: append, shifting the new times by the number of clicks so far."""
    base = sum(len(v) for v in current_clicks_time.values())
    for obj_id, rows in new_clicks.items():
        current_clicks[obj_id].extend(rows)
        current_clicks_time[obj_id].extend([t + base for t in new_click_time[obj_id]])
    return current_clicks, current_clicks_time


def interactive_rounds(forward_mask_fn, labels, raw_coords, num_obj, max_clicks_per_obj,
                       labels_full=None, inverse_map=None):
    """The per-scene loop of Evaluate (eval_multi_obj.py:100-160) for ONE sample, with the model call
    abstracted: forward_mask_fn(click_idx, click_time_idx) -> logits [N, 1+K].  Yields one record per
    round: (current_num_clicks, pred (after the sparse-gt update), iou, click_idx, click_time_idx)."""
    labels_full = labels if labels_full is None else labels_full
    click_idx = {str(k): [] for k in range(num_obj + 1)}
    click_time = {str(k): [] for k in range(num_obj + 1)}
    n_clicks = 0
    records = []
    while n_clicks <= num_obj * max_clicks_per_obj:
        if n_clicks == 0:
            pred = torch.zeros(labels.shape)
        else:
            pred = forward_mask_fn(click_idx, click_time).argmax(-1)
            for obj_id, rows in click_idx.items():
                pred[rows] = int(obj_id)
        full = pred if inverse_map is None else pred[inverse_map]
        iou, _ = mean_iou_scene(full, labels_full)
        records.append({"num_clicks": n_clicks, "pred": pred.clone(), "iou": np.float32(iou),
                        "click_idx": {k: list(v) for k, v in click_idx.items()},
                        "click_time_idx": {k: list(v) for k, v in click_time.items()}})
        new_clicks, _, _, new_time = get_simulated_clicks(pred, labels, raw_coords, n_clicks, training=False)
        if new_clicks is not None:
            click_idx, click_time = extend_clicks(click_idx, click_time, new_clicks, new_time)
        n_clicks += num_obj if n_clicks == 0 else 1
    return records
