"""Interactive multi-object evaluation: the loop of the reference's ``eval_multi_obj.py`` and the
NoC@q / IoU@k tables of ``evaluation/evaluator_MO.py``.

``Evaluate(model, data_loader, args, device)`` takes what the reference's loop takes (a model with
forward_backbone / forward_mask, batches in the reference dataset's collate format, ``args`` with
``output_dir``, ``max_num_clicks``, ``val_list``) and writes the same ``val_results_multi.csv`` rows:

    <instance idx> <scene name without 'scene'> <num_obj> <clicks per object> <mean IoU>

Per round everything stays on the GPU (decoder, label argmax, IoU counts, click simulator); what
crosses to the host is the list of error clusters (a few dozen entries) and 3 x 256 counters.
"""
from __future__ import annotations

import copy
import json
import os

import numpy as np
import torch

from .clicks import argmax_labels, extend_clicks, get_simulated_clicks, mean_iou_scene
from .sparse import SparseTensor


def Evaluate(model, data_loader, args, device, on_round=None):
    """eval_multi_obj.py:76-173.  Returns the results dict of EvaluatorMO when ``args.val_list`` is set
    (the reference computes and prints it), else the path of the CSV.

    ``on_round(sample, current_num_clicks, pred, iou, click_idx, click_time_idx)`` is an optional
    observer (tests use it); it sees the state BEFORE the next clicks are added."""
    model.eval()
    os.makedirs(args.output_dir, exist_ok=True)
    results_file = os.path.join(args.output_dir, "val_results_multi.csv")
    instance_counter = 0
    with open(results_file, "w") as f:
        for batch in data_loader:
            coords, raw_coords, feats, labels, labels_full, inverse_map, click_idx, scene_name, num_obj = batch
            coords = coords.to(device)
            raw_coords = raw_coords.to(device)
            labels = [l.to(device) for l in labels]
            labels_full = [l.to(device) for l in labels_full]
            inverse_map = [(m if torch.is_tensor(m) else torch.as_tensor(np.asarray(m))).to(device)
                           for m in inverse_map]
            data = SparseTensor(coordinates=coords, features=feats, device=device)
            batch_idx = coords[:, 0]
            n_samples = int(batch_idx.max()) + 1
            masks = [batch_idx == i for i in range(n_samples)]
            for sample_clicks in click_idx:               # click ids set null
                for obj_id in sample_clicks:
                    sample_clicks[obj_id] = []
            click_time_idx = copy.deepcopy(click_idx)
            backbone_out = model.forward_backbone(data, raw_coordinates=raw_coords)   # once per scene
            current = 0
            # like the reference, a batch advances in lock-step on the LAST sample's object count
            # (eval_multi_obj.py:114,162-166; val_batch_size is 1 in practice)
            while current <= num_obj[0] * args.max_num_clicks:
                if current:
                    logits = model.forward_mask(*backbone_out, click_idx=click_idx,
                                                click_time_idx=click_time_idx)["pred_masks"]
                for idx in range(n_samples):
                    if current == 0:
                        pred = torch.zeros(labels[idx].shape[0], dtype=torch.int32, device=device)
                    else:
                        pred = argmax_labels(logits[idx], click_idx[idx])     # + sparse-gt update
                    iou, _ = mean_iou_scene(pred, labels_full[idx], inverse_map[idx])
                    f.write(f"{instance_counter + idx} {scene_name[idx].replace('scene', '')} {num_obj[idx]} "
                            f"{current / num_obj[idx]} {iou.cpu().numpy()}\n")
                    if on_round is not None:
                        on_round(idx, current, pred, iou, click_idx[idx], click_time_idx[idx])
                    new_clicks, _, _, new_time = get_simulated_clicks(pred, labels[idx], raw_coords[masks[idx]],
                                                                      current, training=False)
                    if new_clicks is not None:
                        extend_clicks(click_idx[idx], click_time_idx[idx], new_clicks, new_time)
                current += num_obj[n_samples - 1] if current == 0 else 1
            instance_counter += len(num_obj)
    if getattr(args, "val_list", None):
        return EvaluatorMO(args.val_list, results_file, [0.5, 0.65, 0.8, 0.85, 0.9]).eval_results()
    return results_file


class EvaluatorMO:
    """evaluation/evaluator_MO.py:10-139: NoC@q (clicks per object until the scene's mean IoU reaches q,
    capped at the first row with >= 20 clicks) and IoU@k (mean IoU after exactly k clicks per object)."""

    def __init__(self, scene_list_file, result_file, MAX_IOU):
        self.MAX_IOU = MAX_IOU
        with open(scene_list_file) as fh:
            self.dataset_list = json.load(fh)
        self.result_file = result_file

    def _rows(self, dataset_):
        wanted = {k.replace("scene", "").replace("obj_", "") for k in dataset_}
        with open(self.result_file) as fh:
            for line in fh:
                parts = line.rstrip().split(" ")
                if len(parts) < 5:
                    continue
                key = parts[1].replace("scene", "") + "_" + parts[2]
                if key in wanted:
                    yield key, parts[3], parts[4]

    def eval_per_class(self, MAX_IOU=0.8, dataset_=None):
        dataset_ = self.dataset_list if dataset_ is None else dataset_
        first_hit = {}          # object key -> clicks at which it is counted (insertion ordered)
        iou_sum, rows_at = {}, {}
        for key, clicks_s, iou_s in self._rows(dataset_):
            clicks, iou = float(clicks_s), float(iou_s)
            if key not in first_hit and (iou >= MAX_IOU or (clicks >= 20 and iou >= 0)):
                first_hit[key] = clicks
            rows_at[clicks_s] = rows_at.get(clicks_s, 0) + 1
            iou_sum[clicks_s] = iou_sum.get(clicks_s, 0) + iou
        if not first_hit:
            print("no objects to eval")
            return 0
        ordered = list(first_hit.values())
        return ordered, sum(ordered), len(ordered), iou_sum, rows_at

    def eval_results(self):
        noc = {}
        iou_sum = rows_at = None
        for q in self.MAX_IOU:
            _, clicks, objects, iou_sum, rows_at = self.eval_per_class(q, self.dataset_list)
            noc[q] = clicks / objects
        results = {f"NoC@{int(round(100 * q))}": noc[q] for q in (0.5, 0.65, 0.8, 0.85, 0.9)}
        for k in (1, 3, 5, 10, 15):
            results[f"IoU@{k}"] = iou_sum[f"{k}.0"] / rows_at[f"{k}.0"]
        print(results)
        return results
