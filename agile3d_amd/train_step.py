"""One iteration of the reference's training loop on the HIP library (``engine.py:38-150``, SURVEY.md section 8 row
f-2): backbone forward in training mode, random object selection and click simulation with no-grad decoder passes,
decoder forward in training mode, the mask losses with click weights, backward through decoder and backbone,
gradient-norm clip, AdamW.

    stats = train_one_step(model, criterion, optimizer, batch, device, max_norm=0.1)

``batch`` is what ``datasets.collation_fn`` produces.  Random numbers are drawn from ``np.random``, ``torch`` and
``random`` in the reference's order (object count, object permutation, number of click rounds, click order).  With
``torch.distributed`` initialised the gradients are averaged over the ranks before the clip (what DDP does).
"""
from __future__ import annotations

import copy
import os
import random
import sys
import time

import numpy as np
import torch

from .clicks import argmax_labels, argmax_labels_batch, cal_click_loss_weights, extend_clicks, get_simulated_clicks, get_simulated_clicks_batch
from .engine import Scene
from .optim import allreduce_mean_, clip_grad_norm_
from .train_backbone import BackboneTape
from .train_decoder import DecoderTape


def train_one_step(model, criterion, optimizer, batch, device, max_norm: float = 0.1):
    """One iteration of engine.py:38-150 (see the module docstring).  Python's cycle collector is held off for the duration
    of the iteration: the tapes create tens of thousands of container objects, a generation-2 collection in the middle of
    a phase costs 50-100 ms (measured as phases that randomly take 80 ms instead of 2), and the tapes break their own
    reference cycles when they are released at the end."""
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        return _train_one_step(model, criterion, optimizer, batch, device, max_norm)
    finally:
        if gc_was_on:
            gc.enable()


def _train_one_step(model, criterion, optimizer, batch, device, max_norm):
    coords, raw_coords, feats, labels, _, _, click_idx, _scene_name, _num_obj = batch
    timing = os.environ.get("A3D_TRAIN_TIMING")           # phase wall times (device-synchronised) on stderr
    marks = []

    def mark(name):
        if timing:
            torch.cuda.synchronize()
            if timing == "mark":              # one marker dispatch per phase boundary (tools/train_phase_trace.py splits a trace there)
                torch.cuda._sleep(1)
                torch.cuda.synchronize()
            marks.append((name, time.perf_counter()))
    mark("start")
    n_samples = int(coords[:, 0].max()) + 1           # on the host copy when the batch arrives there (collation_fn): no device round trip
    coords = coords.to(device)
    raw_coords = raw_coords.to(device)
    feats = feats.to(device)
    labels = [l.to(device) for l in labels]
    batch_idx = coords[:, 0]
    click_idx = [dict(c) for c in click_idx]

    # ---- backbone, training mode (BatchNorm on the statistics of this batch), engine.py:53
    model.train()
    from .train_backbone import packed_weights_of
    packed_weights_of(model).refresh_all()      # scene-independent: queued before the scene build's host round trip, not behind it
    scene = Scene(coords.to(torch.int32).contiguous())
    bb = BackboneTape(model, scene, feats)
    # the gradient reducer (one small blocking digest all-reduce over the ranks, a dict over the 268 parameters) is made HERE,
    # while the device works through the backbone's launches -- not at the start, where it would sit in front of the first
    # kernel, and not between the decoder's and the backbone's backward, where the host would stall behind the queue
    from .optim import OverlappedAllReduce
    reducer = OverlappedAllReduce(bucket_bytes=int(float(os.environ.get("A3D_DP_BUCKET_MB", "32")) * (1 << 20)),
                                  expected={k: p.numel() for k, p in model.named_parameters() if p.requires_grad},
                                  single_rank=os.environ.get("A3D_DP_SINGLE_RANK", "0") == "1")
    pcd = bb.output
    ranges = scene.batch_ranges
    mark("backbone forward")

    # ---- objects of this iteration, engine.py:55-77
    labels_new, num_objs = [], []
    for idx in range(n_samples):
        sample_labels = labels[idx]
        valid = torch.unique(sample_labels)
        valid = valid[valid != -1]
        max_num_obj = len(valid)
        num_obj = np.random.randint(1, min(10, max_num_obj) + 1)
        obj_idxs = valid[torch.randperm(max_num_obj)[:num_obj].to(valid.device)]
        new = torch.zeros(sample_labels.shape[0], device=device)
        for i, obj_id in enumerate(obj_idxs):
            new[sample_labels == obj_id] = i + 1
            click_idx[idx][str(i + 1)] = []
        click_idx[idx]["0"] = []
        labels_new.append(new)
        num_objs.append(num_obj)          # = the nonzero ids of labels_new[idx] (every chosen object has points)
    click_time_idx = copy.deepcopy(click_idx)
    labels_i32 = [l.to(torch.int32) for l in labels_new]     # once: the click simulator and the criterion read ids (a float -> int
                                                             # conversion per sample and round otherwise)

    # ---- click simulation with the current weights, no gradient (engine.py:82-116)
    num_forward_iters = random.randint(0, 19)
    model.eval()
    eng = model._get_engine()
    eng.refresh_decoder_if_stale()           # only the decoder's packed weights are used here (the previous step's
                                             # optimiser marked everything stale; the backbone program is rebuilt on demand)
    dec_in = eng.decoder_inputs_batch(pcd, raw_coords, ranges)
    pos_enc = dec_in[3][4][0]
    raw_s = [raw_coords[s:e] for (s, e) in ranges]
    fine = timing == "2"                      # A3D_TRAIN_TIMING=2: the click rounds by part (device-synchronised: slower)
    parts = [0.0, 0.0, 0.0, 0.0]

    def lap(i, t0):
        if fine:
            torch.cuda.synchronize()
            parts[i] += time.perf_counter() - t0
        return time.perf_counter()
    for it in range(num_forward_iters + 1):
        t0 = lap(3, time.perf_counter()) if fine else 0.0
        if it:                                 # one batched decoder pass for all samples (5 launches per layer)
            out = eng.forward_mask(*dec_in, click_idx=click_idx, click_time_idx=click_time_idx)
        t0 = lap(0, t0)
        # argmax + "update prediction with sparse gt" (engine.py:96-101) in one kernel instead of 1 + K torch ops; then the
        # samples' error clusters side by side (one host round trip per round, not one per sample), clicks in sample order
        preds = ([torch.zeros(e - s, dtype=torch.int32, device=device) for (s, e) in ranges] if it == 0
                 else argmax_labels_batch(out["pred_masks"], click_idx))
        t0 = lap(1, t0)
        sims = get_simulated_clicks_batch(preds, labels_i32, raw_s, it, training=True, num_objs=num_objs)
        t0 = lap(2, t0)
        for idx, (new_clicks, _, _, new_time) in enumerate(sims):
            if new_clicks is not None:
                click_idx[idx], click_time_idx[idx] = extend_clicks(click_idx[idx], click_time_idx[idx], new_clicks,
                                                                    new_time)
    if fine:
        os.write(2, ("click rounds by part: decoder passes %.1f ms, argmax %.1f ms, clusters + clicks %.1f ms, host between %.1f ms; "
                     "queries at the end %s\n" % (1e3 * parts[0], 1e3 * parts[1], 1e3 * parts[2], 1e3 * parts[3],
                                                    [sum(len(v) for v in c.values()) + 10 for c in click_idx])).encode())
    model.train()
    mark(f"click simulation ({num_forward_iters} decoder rounds)")

    # ---- decoder, training mode: ONE tape for the batch (agile3d.py:192 loops over the samples, which only share the
    # weights: row-wise layers run once over all samples' rows, attention and the mask head per sample on row ranges)
    tape = DecoderTape(model, [pcd[s:e] for (s, e) in ranges], [pos_enc[i] for i in range(len(ranges))], click_idx,
                       click_time_idx)
    n_layers = len(tape.logits)
    outputs = {"pred_masks": list(tape.logits[-1]),
               "aux_outputs": [{"pred_masks": list(tape.logits[l])} for l in range(n_layers - 1)]}

    mark("decoder forward")
    # ---- losses (engine.py:124-128) and their gradient with respect to every level's logits
    # one pass of the loss kernel per level and sample gives the values AND the gradient; one host round trip for all values
    click_weights = cal_click_loss_weights(batch_idx, raw_coords, None, click_idx, ranges=ranges)
    targets = labels_i32
    loss_dict, gl = criterion.forward_and_grad(outputs, targets, click_weights)
    loss_keys = list(loss_dict)
    loss_vals = dict(zip(loss_keys, torch.stack([loss_dict[k] for k in loss_keys]).tolist()))
    total = sum(loss_vals[k] * criterion.weight_dict[k] for k in loss_keys if k in criterion.weight_dict)
    if not np.isfinite(total):
        raise FloatingPointError(f"Loss is {total}, stopping training")

    mark("losses")
    # ---- backward: decoders, then the backbone through d(pcd_features)
    n_s = len(ranges)
    dl = [[gl["aux_outputs"][l][i] for i in range(n_s)] for l in range(n_layers - 1)] + [[gl["pred_masks"][i] for i in range(n_s)]]
    grads, d_pcd = tape.backward(dl)         # rows of the samples are contiguous and in order: d_pcd is [N_total, 128]
    grads = dict(grads)
    mark("decoder backward")
    # ---- data-parallel average (engine.py runs under DDP: main.py:115-127), overlapped with the backbone backward: the
    # decoder's gradients are final here, the U-Net's become final from the head down to the stem
    for k in sorted(grads):
        reducer.add(k, grads[k])
    reducer.flush()
    grads.update(bb.backward(d_pcd, on_grad=reducer.add if reducer.active else None))
    mark("backbone backward")
    reducer.finish(grads)
    tape.release()                       # break the tape <-> closure cycles: the activations are freed NOW, the next
    bb.release()                         # iteration reuses their blocks instead of growing the allocator's pools
    del tape, bb

    # ---- clip, AdamW (engine.py:143-150)
    norm, coef = clip_grad_norm_(grads, max_norm)
    optimizer.step(grads, coef)
    eng.mark_stale()
    mark("clip + AdamW")
    stats = {"loss": total, "grad_norm": norm, "loss_dict": loss_vals,
             "clicks": [sum(len(v) for v in c.values()) for c in click_idx], "click_rounds": num_forward_iters}
    if timing:
        import sys
        stats["phases_ms"] = {n.split(" (")[0]: 1e3 * (t - marks[i][1]) for i, (n, t) in enumerate(marks[1:])}
        if timing not in ("quiet", "mark"):
            print("train_one_step: " + ", ".join(f"{n} {1e3 * (t - marks[i][1]):.1f} ms" for i, (n, t) in enumerate(marks[1:])),
                  file=sys.stderr)
    return stats


def train_one_step_api(model, criterion, optimizer, batch, device, max_norm: float = 0.1):
    """The same iteration written the way the reference's ``train_one_epoch`` body is (engine.py:38-150), against
    nothing but the model's public interface and torch: ``model.train()`` -> ``forward_backbone`` -> object sampling
    -> ``model.eval()`` + ``no_grad`` click rounds through ``forward_mask`` -> ``model.train()`` -> ``forward_mask`` ->
    ``criterion`` -> weighted sum -> ``optimizer.zero_grad(); losses.backward(); clip_grad_norm_; optimizer.step()``
    with a ``torch.optim`` optimiser.  Forward and backward arithmetic run in the HIP library (agile3d_amd/autograd.py
    ties the tapes into torch's graph).  Same random-number consumption as ``train_one_step``; with
    ``torch.distributed`` initialised the ``.grad`` tensors are averaged over the ranks before the clip."""
    from .sparse import SparseTensor
    coords, raw_coords, feats, labels, _, _, click_idx, _scene_name, _num_obj = batch
    coords = coords.to(device)
    raw_coords = raw_coords.to(device)
    feats = feats.to(device)
    labels = [l.to(device) for l in labels]
    click_idx = [dict(c) for c in click_idx]
    model.train()
    criterion.train()
    data = SparseTensor(coordinates=coords, features=feats, device=device)
    pcd_features, aux, coordinates, pos_encodings_pcd = model.forward_backbone(data, raw_coordinates=raw_coords)

    batch_idx = coords[:, 0]
    n_samples = int(batch_idx.max()) + 1
    labels_new = []
    for idx in range(n_samples):
        sample_labels = labels[idx]
        valid = torch.unique(sample_labels)
        valid = valid[valid != -1]
        max_num_obj = len(valid)
        num_obj = np.random.randint(1, min(10, max_num_obj) + 1)
        obj_idxs = valid[torch.randperm(max_num_obj)[:num_obj].to(valid.device)]
        new = torch.zeros(sample_labels.shape[0], device=device)
        for i, obj_id in enumerate(obj_idxs):
            new[sample_labels == obj_id] = i + 1
            click_idx[idx][str(i + 1)] = []
        click_idx[idx]["0"] = []
        labels_new.append(new)
    click_time_idx = copy.deepcopy(click_idx)

    num_forward_iters = random.randint(0, 19)
    with torch.no_grad():
        model.eval()
        masks = [batch_idx == i for i in range(n_samples)]
        for it in range(num_forward_iters + 1):
            if it:
                out = model.forward_mask(pcd_features, aux, coordinates, pos_encodings_pcd, click_idx=click_idx,
                                         click_time_idx=click_time_idx)
            for idx in range(n_samples):
                if it == 0:
                    pred = torch.zeros(int(masks[idx].sum()), device=device)
                else:
                    pred = out["pred_masks"][idx].argmax(-1)
                    for obj_id, cids in click_idx[idx].items():
                        pred[cids] = int(obj_id)
                new_clicks, _, _, new_time = get_simulated_clicks(pred, labels_new[idx], raw_coords[masks[idx]], it,
                                                                  training=True)
                if new_clicks is not None:
                    click_idx[idx], click_time_idx[idx] = extend_clicks(click_idx[idx], click_time_idx[idx], new_clicks,
                                                                        new_time)
        model.train()

    outputs = model.forward_mask(pcd_features, aux, coordinates, pos_encodings_pcd, click_idx=click_idx,
                                 click_time_idx=click_time_idx)
    click_weights = cal_click_loss_weights(batch_idx, raw_coords, torch.cat(labels_new), click_idx)
    loss_dict = criterion(outputs, labels_new, click_weights)
    weight_dict = criterion.weight_dict
    losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
    loss_value = float(losses.detach())
    if not np.isfinite(loss_value):
        raise FloatingPointError(f"Loss is {loss_value}, stopping training")
    optimizer.zero_grad()
    losses.backward()
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        allreduce_mean_(grads)
    grad_total_norm = None
    if max_norm > 0:
        grad_total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    optimizer.step()
    return {"loss": loss_value, "grad_norm": None if grad_total_norm is None else float(grad_total_norm),
            "loss_dict": {k: float(v.detach()) for k, v in loss_dict.items()},
            "clicks": [sum(len(v) for v in c.values()) for c in click_idx]}


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch: int, train_total_iter: int = 0,
                    max_norm: float = 0.0, print_freq: int = 10, log=print):
    """engine.py:26-179 around ``train_one_step``: one pass over ``data_loader``; returns (averaged statistics,
    train_total_iter) like the reference (its wandb / MetricLogger bookkeeping reduced to running means)."""
    sums, n = {}, 0
    for i, batch in enumerate(data_loader):
        st = train_one_step(model, criterion, optimizer, batch, device, max_norm)
        train_total_iter += 1
        n += 1
        for k, v in {"loss": st["loss"], "grad_norm": st["grad_norm"], **st["loss_dict"]}.items():
            sums[k] = sums.get(k, 0.0) + float(v)
        if log is not None and i % print_freq == 0:
            log(f"Epoch: [{epoch}] [{i}/{len(data_loader)}] loss {st['loss']:.4f} grad_norm {st['grad_norm']:.3f} "
                f"lr {optimizer.lr:.6f}")
    return {k: v / max(n, 1) for k, v in sums.items()}, train_total_iter


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones) as main.py:127 uses it (gamma 0.1), for
    ``agile3d_amd.optim.AdamW``: call ``step()`` once per epoch."""

    def __init__(self, optimizer, milestones, gamma=0.1):
        self.optimizer, self.milestones, self.gamma = optimizer, sorted(milestones), gamma
        self.base_lr, self.last_epoch = optimizer.lr, 0

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr = self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= self.last_epoch)

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lr": self.base_lr, "milestones": self.milestones, "gamma": self.gamma}

    def load_state_dict(self, sd, keep_base_lr=False):
        """``keep_base_lr``: restore the position in the schedule (epoch, milestones, gamma) but keep the base learning
        rate of the RUNNING configuration -- what resuming with a changed --lr means (main.py:131-152 keeps the new
        lr of the param groups).  Accepts this class's own state and torch.optim.lr_scheduler.MultiStepLR's
        (``base_lrs`` list, ``milestones`` as a Counter)."""
        ms = sd["milestones"]
        self.milestones = sorted(ms.elements()) if hasattr(ms, "elements") else sorted(ms)
        self.last_epoch, self.gamma = sd["last_epoch"], sd["gamma"]
        if not keep_base_lr:
            if "base_lr" in sd:
                self.base_lr = sd["base_lr"]
            elif sd.get("base_lrs"):
                self.base_lr = sd["base_lrs"][0]
        self.optimizer.lr = self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= self.last_epoch)


def save_checkpoint(path, model, optimizer, lr_scheduler, epoch, args=None):
    """The reference's checkpoint file (main.py:190-201): {'model', 'optimizer', 'lr_scheduler', 'epoch', 'args'}; the
    optimiser state is in torch.optim.AdamW's own layout."""
    torch.save({"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "optimizer": optimizer.state_dict(),
                "lr_scheduler": lr_scheduler.state_dict() if lr_scheduler is not None else None, "epoch": epoch,
                "args": args}, path)


def load_checkpoint(path, model, optimizer=None, lr_scheduler=None, reference_schedule=False):
    """Resume like main.py:131-152: weights with strict=False, then optimiser state and epoch when the file has them
    (the learning rate of the running configuration is kept, as the reference does).  Returns the epoch to start from.

    Scheduler: the reference does NOT load the scheduler's state (the line is commented out, main.py:146) -- it calls
    ``lr_scheduler.step(lr_scheduler.last_epoch)`` on the fresh scheduler, so after a resume past ``lr_drop`` it trains
    at the undropped rate again.  Default here (a deliberate deviation): the position in the schedule is restored from
    the file, so a resumed run continues the uninterrupted run bit for bit (tests/test_gpu_backward.py).
    ``reference_schedule=True`` gives the reference's behaviour: the scheduler is left at epoch 0 of the running lr."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    missing, unexpected = model.load_state_dict(ck["model"], strict=False)
    unexpected = [k for k in unexpected if not (k.endswith("total_params") or k.endswith("total_ops"))]
    if missing:
        print("Missing Keys: {}".format(missing))
    if unexpected:
        print("Unexpected Keys: {}".format(unexpected))
    start = 0
    if optimizer is not None and "optimizer" in ck and "epoch" in ck and ck["optimizer"] is not None:
        lr = optimizer.lr
        optimizer.load_state_dict(ck["optimizer"])
        optimizer.lr = lr
        if lr_scheduler is not None:
            lr_scheduler.base_lr = lr                                   # a changed --lr stays in force after the resume
            if ck.get("lr_scheduler") is not None and not reference_schedule:
                lr_scheduler.load_state_dict(ck["lr_scheduler"], keep_base_lr=True)
        start = ck["epoch"] + 1
    return start


def evaluate(model, criterion, data_loader, args, epoch, device):
    """engine.py:182-300: the validation pass of the training script -- the interactive protocol of ``Evaluate`` with
    the losses of every click round on top; writes ``val_results_epoch_<epoch>.csv`` into ``args.valResults_dir`` and
    returns the averaged statistics + the NoC / IoU@k table.  No gradients; inference kernels only."""
    import os as _os

    from .clicks import argmax_labels, mean_iou_scene
    from .evaluate import EvaluatorMO
    from .sparse import SparseTensor
    model.eval()
    criterion.eval()
    _os.makedirs(args.valResults_dir, exist_ok=True)
    results_file = _os.path.join(args.valResults_dir, "val_results_epoch_" + str(epoch) + ".csv")
    sums, rounds, instance_counter = {}, 0, 0
    with open(results_file, "w") as f:
        for batch in data_loader:
            coords, raw_coords, feats, labels, labels_full, inverse_map, click_idx, scene_name, num_obj = batch
            coords, raw_coords = coords.to(device), raw_coords.to(device)
            labels = [l.to(device) for l in labels]
            labels_full = [l.to(device=device, dtype=torch.int32) for l in labels_full]
            labels_i32 = [l.to(torch.int32) for l in labels]     # for the IoU / click kernels: converted once, not per round
            inverse_map = [(m if torch.is_tensor(m) else torch.as_tensor(np.asarray(m))).to(device) for m in inverse_map]
            batch_idx = coords[:, 0]
            n_samples = int(batch_idx.max()) + 1
            masks = [batch_idx == i for i in range(n_samples)]
            for c in click_idx:
                for obj_id in c:
                    c[obj_id] = []
            click_time_idx = copy.deepcopy(click_idx)
            backbone_out = model.forward_backbone(SparseTensor(coordinates=coords, features=feats, device=device),
                                                  raw_coordinates=raw_coords)
            current, max_clicks = 0, num_obj[0] * args.max_num_clicks
            while current <= max_clicks:
                if current:
                    outputs = model.forward_mask(*backbone_out, click_idx=click_idx, click_time_idx=click_time_idx)
                    cw = cal_click_loss_weights(batch_idx, raw_coords, torch.cat(labels), click_idx)
                    loss_dict = criterion(outputs, labels, cw)
                    scaled = {k: float(v) * criterion.weight_dict[k] for k, v in loss_dict.items() if k in criterion.weight_dict}
                    for k, v in {"loss": sum(scaled.values()), **scaled, **{k + "_unscaled": float(v) for k, v in loss_dict.items()}}.items():
                        sums[k] = sums.get(k, 0.0) + v
                miou = 0.0
                for idx in range(n_samples):
                    if current == 0:
                        pred = torch.zeros(labels[idx].shape[0], dtype=torch.int32, device=device)
                    else:
                        pred = argmax_labels(outputs["pred_masks"][idx], click_idx[idx])        # + sparse-gt update
                        miou += float(mean_iou_scene(pred, labels_i32[idx])[0])
                    iou, _ = mean_iou_scene(pred, labels_full[idx], inverse_map[idx])
                    f.write(f"{instance_counter + idx} {scene_name[idx].replace('scene', '')} {num_obj[idx]} "
                            f"{current / num_obj[idx]} {iou.cpu().numpy()}\n")
                    new_clicks, _, _, new_time = get_simulated_clicks(pred, labels_i32[idx], raw_coords[masks[idx]], current,
                                                                      training=False)
                    if new_clicks is not None:
                        extend_clicks(click_idx[idx], click_time_idx[idx], new_clicks, new_time)
                if current:
                    sums["mIoU"] = sums.get("mIoU", 0.0) + miou / n_samples
                    rounds += 1
                current += num_obj[n_samples - 1] if current == 0 else 1
            instance_counter += len(num_obj)
    stats = {k: v / max(rounds, 1) for k, v in sums.items()}
    stats.update(EvaluatorMO(args.val_list, results_file, [0.5, 0.65, 0.8, 0.85, 0.9]).eval_results())
    return stats
