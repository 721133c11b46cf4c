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

from .clicks import argmax_labels_batch, extend_clicks, mean_iou_and_clusters_batch, pick_clicks_batch
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
            # object ids as int32 ONCE per scene: the IoU / click kernels read int32 and would convert per round otherwise
            labels = [l.to(device=device, dtype=torch.int32) for l in labels]
            labels_full = [l.to(device=device, dtype=torch.int32) for l in labels_full]
            inverse_map = [(m if torch.is_tensor(m) else torch.as_tensor(np.asarray(m))).to(device)
                           for m in inverse_map]
            data = SparseTensor(coordinates=coords, features=feats, device=device)
            batch_idx = coords[:, 0]
            n_samples = int(batch_idx.max()) + 1
            raw_s = [raw_coords[batch_idx == i] for i in range(n_samples)]
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
                # the samples of a batch side by side, IoU counts and error clusters behind ONE host round trip (the clusters
                # do not depend on the IoU; their kernels overlap on side streams); clicks are picked in sample order
                preds = ([torch.zeros(labels[idx].shape[0], dtype=torch.int32, device=device) for idx in range(n_samples)]
                         if current == 0 else argmax_labels_batch(logits, click_idx))       # + sparse-gt update, all samples in two launches
                ious, clusters = mean_iou_and_clusters_batch(preds, labels_full, inverse_map, labels, raw_s)
                for idx in range(n_samples):
                    iou = ious[idx][0]
                    f.write(f"{instance_counter + idx} {scene_name[idx].replace('scene', '')} {num_obj[idx]} "
                            f"{current / num_obj[idx]} {iou.cpu().numpy()}\n")
                    if on_round is not None:
                        on_round(idx, current, preds[idx], iou, click_idx[idx], click_time_idx[idx])
                sims = pick_clicks_batch(clusters, labels, raw_s, current, training=False)
                for idx, (new_clicks, _, _, new_time) in enumerate(sims):
                    if new_clicks is not None:
                        extend_clicks(click_idx[idx], click_time_idx[idx], new_clicks, new_time)
                current += num_obj[n_samples - 1] if current == 0 else 1
            instance_counter += len(num_obj)
    if getattr(args, "val_list", None):
        return EvaluatorMO(args.val_list, results_file, [0.5, 0.65, 0.8, 0.85, 0.9]).eval_results()
    return results_file


def EvaluateSingle(model, data_loader, args, device, on_round=None):
    """eval_single_obj.py:76-173: the single-object protocol -- one target object per sample (labels are 0/1), click
    dicts {'0': [], '1': []}, one new click per round up to ``args.max_num_clicks``; rows of
    ``val_results_single.csv`` are ``<instance idx> <scene> <object id> <num clicks> <IoU>``.  Returns the results
    dict of EvaluatorSO when ``args.val_list`` and ``args.val_list_classes`` are set, else the CSV path."""
    model.eval()
    os.makedirs(args.output_dir, exist_ok=True)
    results_file = os.path.join(args.output_dir, "val_results_single.csv")
    instance_counter = 0
    with open(results_file, "w") as f:
        for batch in data_loader:
            coords, raw_coords, feats, labels, labels_full, inverse_map, _click_idx, scene_name, object_id = batch
            coords = coords.to(device)
            raw_coords = raw_coords.to(device)
            # object ids as int32 ONCE per scene: the IoU / click kernels read int32 and would convert per round otherwise
            labels = [l.to(device=device, dtype=torch.int32) for l in labels]
            labels_full = [l.to(device=device, dtype=torch.int32) for l in labels_full]
            inverse_map = [(m if torch.is_tensor(m) else torch.as_tensor(np.asarray(m))).to(device)
                           for m in inverse_map]
            data = SparseTensor(coordinates=coords, features=feats, device=device)
            batch_idx = coords[:, 0]
            n_samples = int(batch_idx.max()) + 1
            raw_s = [raw_coords[batch_idx == i] for i in range(n_samples)]
            click_idx = [{"0": [], "1": []} for _ in range(n_samples)]
            click_time_idx = copy.deepcopy(click_idx)
            backbone_out = model.forward_backbone(data, raw_coordinates=raw_coords)
            for current in range(args.max_num_clicks + 1):
                if current:
                    logits = model.forward_mask(*backbone_out, click_idx=click_idx,
                                                click_time_idx=click_time_idx)["pred_masks"]
                preds = ([torch.zeros(labels[idx].shape[0], dtype=torch.int32, device=device) for idx in range(n_samples)]
                         if current == 0 else argmax_labels_batch(logits, click_idx))
                # IoU counts and error clusters of all samples behind ONE host synchronisation (the clusters do not depend
                # on the IoU; the clicks are picked afterwards, in sample order, so the random stream is consumed as before)
                ious, clusters = mean_iou_and_clusters_batch(preds, labels_full, inverse_map, labels, raw_s)
                for idx in range(n_samples):
                    iou = ious[idx][0]
                    f.write(f"{instance_counter + idx} {scene_name[idx].replace('scene', '')} {object_id[idx]} "
                            f"{current} {iou.cpu().numpy()}\n")
                    if on_round is not None:
                        on_round(idx, current, preds[idx], iou, click_idx[idx], click_time_idx[idx])
                sims = pick_clicks_batch(clusters, labels, raw_s, current, training=False)
                for idx, (new_clicks, _, _, new_time) in enumerate(sims):
                    if new_clicks is not None:
                        extend_clicks(click_idx[idx], click_time_idx[idx], new_clicks, new_time)
            instance_counter += len(object_id)
    if getattr(args, "val_list", None) and getattr(args, "val_list_classes", None):
        return EvaluatorSO(getattr(args, "dataset", None), args.val_list, args.val_list_classes, results_file,
                           [0.5, 0.65, 0.8, 0.85, 0.9], label_all=getattr(args, "label_all", None)).eval_results()
    return results_file


class EvaluatorSO:
    """evaluation/evaluator_SO.py:10-155: NoC@q / IoU@k of the single-object protocol, accumulated class by class
    over ``label_all`` (the reference takes it from ``evaluation/labels.py[dataset]``, a table of class names that is
    data of the benchmark, not code: pass it in; default = the classes present in the class list file).  Objects are
    rows ``(scene, object id)`` of ``object_list_file`` (.npy) with their class in ``object_classes_list_file``."""

    def __init__(self, dataset, object_list_file, object_classes_list_file, result_file, MAX_IOU, label_all=None):
        self.dataset = dataset
        self.MAX_IOU = MAX_IOU
        self.dataset_list = np.load(object_list_file)
        self.dataset_classes = np.loadtxt(object_classes_list_file, dtype=str)
        self.label_all = list(label_all) if label_all is not None else sorted(set(self.dataset_classes.tolist()))
        self.result_file = result_file

    def eval_per_class(self, label=None, MAX_IOU=0.8, dataset_=None, dataset_classes=None,
                       exclude_classes=("wall", "ceiling", "floor", "unlabelled", "unlabeled")):
        dataset_ = self.dataset_list if dataset_ is None else dataset_
        dataset_classes = self.dataset_classes if dataset_classes is None else dataset_classes
        if exclude_classes:
            keep = np.isin(dataset_classes, list(exclude_classes), invert=True)
            dataset_, dataset_classes = dataset_[keep], dataset_classes[keep]
        if label:
            dataset_ = dataset_[dataset_classes == label]
        wanted = {row[0].replace("scene", "") + "_" + row[1] for row in dataset_}
        first_hit, iou_sum, rows_at = {}, {}, {}
        with open(self.result_file) as fh:
            for line in fh:
                parts = line.rstrip().split(" ")
                if len(parts) < 5:
                    continue
                key = parts[1].replace("scene", "") + "_" + parts[2]
                if key not in wanted:
                    continue
                clicks_s, iou = parts[3], float(parts[4])
                if key not in first_hit and (iou >= MAX_IOU or (int(clicks_s) >= 20 and iou >= 0)):
                    first_hit[key] = float(clicks_s)
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
            clicks = objects = 0
            iou_sum, rows_at = {}, {}
            for label in set(self.label_all):
                _, c, o, isum, rows = self.eval_per_class(label, q, self.dataset_list, self.dataset_classes,
                                                          exclude_classes=None)
                clicks, objects = clicks + c, objects + o
                for k in isum:
                    iou_sum[k] = iou_sum.get(k, 0) + isum[k]
                    rows_at[k] = rows_at.get(k, 0) + rows[k]
            noc[q] = clicks / objects
        results = {f"NoC@{int(round(100 * q))}": noc[q] for q in (0.5, 0.65, 0.8, 0.85, 0.9)}
        for k in (1, 2, 3, 5, 10, 15):
            results[f"IoU@{k}"] = iou_sum[str(k)] / rows_at[str(k)]
        print(results)
        return results


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
