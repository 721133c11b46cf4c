"""Host mirror of the reference's ``utils/seg.py`` for the interactive loop, on the HIP library.

Same function names, argument meaning and return values as the reference (file:line cited per
function); the arithmetic runs in ``libagile3d_hip.so`` (``csrc/clicks.hip``): one exact
nearest-outside-point pass for all error clusters instead of one ``torch.cdist`` matrix per cluster.
There is no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes as C
import random

import numpy as np
import torch

from . import lib as L

MAX_CLUSTERS = 1024


def _i32(t: torch.Tensor) -> torch.Tensor:
    """labels / predictions arrive as float, long or int tensors in the reference; ids are small ints."""
    if not t.is_cuda:
        raise RuntimeError("agile3d_amd.clicks runs on the GPU only (no CPU fallback); got a CPU tensor")
    return t.to(torch.int32).contiguous()


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _host_i32(values):
    a = np.ascontiguousarray(values, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def argmax_labels(logits: torch.Tensor, click_idx: dict | None = None) -> torch.Tensor:
    """``p.argmax(-1)`` followed by "update prediction with sparse gt" (eval_multi_obj.py:119-134):
    int32 labels [N]; rows listed in ``click_idx`` are overwritten with their object id, dict order."""
    lib = L.load()
    logits = logits.contiguous()
    if logits.dtype != torch.float32 or not logits.is_cuda or logits.dim() != 2:
        raise RuntimeError("argmax_labels: logits must be a CUDA float32 [N, 1+K] tensor")
    rows, objs = [], []
    for obj_id, cids in (click_idx or {}).items():
        rows += [int(c) for c in cids]
        objs += [int(obj_id)] * len(cids)
    r, rp = _host_i32(rows)
    o, op = _host_i32(objs)
    pred = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
    L.check(lib.a3d_argmax_labels(logits.data_ptr(), logits.shape[0], logits.shape[1], rp, op, len(rows),
                                  pred.data_ptr(), _stream(logits)), "a3d_argmax_labels")
    return pred


_MAX_ROUND_SAMPLES = 64     # include/agile3d_hip.h: A3D_MAX_ROUND_SAMPLES


def argmax_labels_batch(logits_list, click_idx_list=None):
    """``argmax_labels`` of every sample of a round in two launches (``a3d_argmax_labels_batch``; the per-sample calls were two
    launches each: 32 per lock-step evaluation round of 16 scenes).  Same results, in sample order."""
    lib = L.load()
    ns = len(logits_list)
    if ns == 0:
        return []
    if ns > _MAX_ROUND_SAMPLES:
        return [argmax_labels(lg, None if click_idx_list is None else click_idx_list[i]) for i, lg in enumerate(logits_list)]
    dev = logits_list[0].device
    arr = (L.ArgmaxSample * ns)()
    keep, preds = [], []
    for i, lg in enumerate(logits_list):
        lg = lg.contiguous()
        if lg.dtype != torch.float32 or not lg.is_cuda or lg.dim() != 2:
            raise RuntimeError("argmax_labels: logits must be a CUDA float32 [N, 1+K] tensor")
        rows, objs = [], []
        for obj_id, cids in ((click_idx_list[i] if click_idx_list is not None else None) or {}).items():
            rows.extend(map(int, cids))
            objs.extend([int(obj_id)] * len(cids))
        r, rp = _host_i32(rows)
        o, op = _host_i32(objs)
        pred = torch.empty(lg.shape[0], dtype=torch.int32, device=dev)
        sp = arr[i]
        sp.logits_dev, sp.n, sp.n_classes, sp.n_clicks = lg.data_ptr(), lg.shape[0], lg.shape[1], len(rows)
        sp.click_row, sp.click_obj, sp.pred_dev = rp, op, pred.data_ptr()
        keep.append((lg, r, o))
        preds.append(pred)
    key = (dev.index, "argmax_ws", ns)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = _ws_cache[key] = torch.empty(lib.a3d_argmax_labels_batch_workspace_bytes(ns), dtype=torch.uint8, device=dev)
    L.check(lib.a3d_argmax_labels_batch(C.cast(arr, C.c_void_p), ns, ws.data_ptr(), ws.numel(), _stream(logits_list[0])),
            "a3d_argmax_labels_batch")
    return preds


def iou_counts(pred, labels, inverse_map=None, n_ids: int | None = None) -> np.ndarray:
    """int64 [3][n_ids]: |pred==id & label==id|, |pred==id|, |label==id| with pred read through
    ``inverse_map`` (voxel -> full-resolution points, eval_multi_obj.py:138) when given."""
    lib = L.load()
    p, l = _i32(pred), _i32(labels)
    inv = None
    if inverse_map is not None:
        inv = inverse_map.to(device=p.device, dtype=torch.int64).contiguous()
        if inv.numel() != l.numel():
            raise RuntimeError("iou_counts: inverse_map and labels differ in length")
    elif p.numel() != l.numel():
        raise RuntimeError("iou_counts: pred and labels differ in length")
    n_ids = 256 if n_ids is None else n_ids
    counts = torch.empty(3 * n_ids + 1, dtype=torch.int64, device=p.device)
    L.check(lib.a3d_iou_counts(p.data_ptr(), p.numel(), inv.data_ptr() if inv is not None else None, l.data_ptr(),
                               l.numel(), n_ids, counts.data_ptr(), _stream(p)), "a3d_iou_counts")
    host = counts.cpu().numpy()
    if host[-1]:
        raise RuntimeError("iou_counts: inverse_map holds rows outside the prediction")
    return host[:-1].reshape(3, n_ids)


def _launch_iou_counts(preds, labels, inverse_maps, n_ids, counts):
    """IoU counts of all samples into ``counts`` [ns][3 n_ids + 1] on the current stream: ONE launch + one clear
    (``a3d_iou_counts_batch``; a launch + a clear per sample before).  Returns the tensors the launch reads."""
    lib = L.load()
    dev = counts.device
    keep = []
    for i, (pr, lb) in enumerate(zip(preds, labels)):
        p, l = _i32(pr), _i32(lb)
        inv = None
        if inverse_maps is not None and inverse_maps[i] is not None:
            inv = inverse_maps[i].to(device=dev, dtype=torch.int64).contiguous()
            if inv.numel() != l.numel():
                raise RuntimeError("iou_counts: inverse_map and labels differ in length")
        elif p.numel() != l.numel():
            raise RuntimeError("iou_counts: pred and labels differ in length")
        keep.append((p, l, inv))
    for c0 in range(0, len(keep), _MAX_ROUND_SAMPLES):
        part = keep[c0:c0 + _MAX_ROUND_SAMPLES]
        arr = (L.IouSample * len(part))()
        for k, (p, l, inv) in enumerate(part):
            sp = arr[k]
            sp.pred_dev, sp.n_pred, sp.inverse_map_dev = p.data_ptr(), p.numel(), inv.data_ptr() if inv is not None else None
            sp.labels_dev, sp.n_full = l.data_ptr(), l.numel()
        L.check(lib.a3d_iou_counts_batch(C.cast(arr, C.c_void_p), len(part), n_ids, counts[c0].data_ptr(), _stream(counts)),
                "a3d_iou_counts_batch")
    return keep


def iou_counts_batch(preds, labels, inverse_maps=None, n_ids: int = 256):
    """``iou_counts`` of several samples with one launch and one device-to-host copy: list of int64 [3][n_ids] arrays."""
    if not preds:
        return []
    dev = preds[0].device
    counts = torch.empty(len(preds), 3 * n_ids + 1, dtype=torch.int64, device=dev)
    keep = _launch_iou_counts(preds, labels, inverse_maps, n_ids, counts)
    host = counts.cpu().numpy()
    if host[:, -1].any():
        raise RuntimeError("iou_counts: inverse_map holds rows outside the prediction")
    return [host[i, :-1].reshape(3, n_ids) for i in range(len(preds))]


def _mean_iou_from_counts(c):
    ids = [i for i in range(1, c.shape[1]) if c[2, i] > 0]
    total = np.float32(0.0)
    per_obj = {}
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in ids:
            v = np.float32(c[0, i]) / np.float32(c[1, i] + c[2, i] - c[0, i])
            per_obj[i] = float(v)
            total = np.float32(total + v)
        total = np.float32(total / np.float32(len(ids))) if ids else np.float32(np.nan)
    return torch.tensor(total, dtype=torch.float32), per_obj


def mean_iou_scene_batch(preds, labels, inverse_maps=None):
    """``mean_iou_scene`` per sample of a batch, one host round trip for all of them."""
    return [_mean_iou_from_counts(c) for c in iou_counts_batch(preds, labels, inverse_maps)]


def mean_iou_scene(pred, labels, inverse_map=None):
    """utils/seg.py:44-59.  Returns (mean IoU as a 0-d float32 tensor, {object id: IoU}); the fp32
    arithmetic (int counts -> fp32 divide -> sequential fp32 sum) is the reference's."""
    return _mean_iou_from_counts(iou_counts(pred, labels, inverse_map))


_ws_cache: dict = {}
_REC = np.dtype([("cluster_id", "<i4"), ("row", "<i4"), ("label", "<i4"), ("pred", "<i4"), ("error_size", "<f4")])
_OUT_BYTES = MAX_CLUSTERS * C.sizeof(L.ClickCluster) + 16


def _cluster_buffers(device, n, slot=0):
    """Workspace + result buffer of one sample in flight (slot = its place in a batch), grown on demand."""
    key = (device.index, slot)
    ws = _ws_cache.get(key)
    need = L.load().a3d_click_workspace_bytes(n)
    if ws is None or ws[0].numel() < need:
        ws = _ws_cache[key] = (torch.empty(need, dtype=torch.uint8, device=device),
                               torch.empty(_OUT_BYTES, dtype=torch.uint8, device=device),
                               torch.empty(_OUT_BYTES, dtype=torch.uint8).pin_memory())
    return ws


def _launch_clusters(p, l, xyz, work, out):
    n = p.numel()
    if xyz.shape != (n, 3) or l.numel() != n:
        raise RuntimeError("error_clusters: pred [N], labels [N], coords [N,3] expected")
    L.check(L.load().a3d_click_clusters(xyz.data_ptr(), p.data_ptr(), l.data_ptr(), n, out.data_ptr(), MAX_CLUSTERS,
                                        out.data_ptr() + MAX_CLUSTERS * C.sizeof(L.ClickCluster), work.data_ptr(),
                                        work.numel(), _stream(p)), "a3d_click_clusters")


def _parse_clusters(host: np.ndarray):
    count = int(host[MAX_CLUSTERS * C.sizeof(L.ClickCluster):][:4].view(np.int32)[0])
    if count < 0:
        raise RuntimeError("error_clusters: labels / predictions must be object ids in 0..255")
    if count > MAX_CLUSTERS:
        raise RuntimeError(f"error_clusters: {count} error clusters > {MAX_CLUSTERS}")
    recs = np.frombuffer(host[:count * C.sizeof(L.ClickCluster)].tobytes(), dtype=_REC)
    if not np.isfinite(recs["error_size"]).all():
        raise RuntimeError("error cluster covers the whole sample (the reference fails here too)")
    # one conversion of all records to Python scalars (field-by-field access to numpy records cost 25 us per sample and round
    # between the device's cluster search and the next decoder pass)
    names = recs.dtype.names
    return [dict(zip(names, t)) for t in recs.tolist()]


def error_clusters(pred, labels, coords):
    """Per error cluster (ascending cluster id = 96*label + 11*pred, utils/seg.py:186): the point
    farthest from everything outside the cluster and that distance.  List of dicts
    {cluster_id, row, label, pred, error_size}."""
    p, l = _i32(pred), _i32(labels)
    xyz = coords.to(torch.float32).contiguous()
    if p.numel() == 0:
        return []
    work, out, _ = _cluster_buffers(p.device, p.numel())
    _launch_clusters(p, l, xyz, work, out)
    return _parse_clusters(out.cpu().numpy())


_MAX_CLICK_BATCH = 64      # csrc/clicks.hip: kMaxClickBatch
_ORDER_FROM = 20000        # csrc/clicks.hip: kCoarseFrom -- smaller samples run no first bounding stage
_order_cache: "dict" = {}  # (device, data_ptr, n) -> (order, inv) of a coordinate tensor, most recent last


def _spatial_order(xyz):
    """Morton order of a sample's coordinates and its inverse (``a3d_click_spatial_order``), cached per coordinate tensor: a
    scene's coordinates do not change over its click rounds (eval_multi_obj.py:112-166, engine.py:83-116).  The cache is
    keyed on the tensor's address and length; a hit on recycled memory hands the search a stale permutation, which costs it
    tightness and never exactness (csrc/clicks.hip: any permutation gives valid bounds)."""
    n = xyz.shape[0]
    if n < _ORDER_FROM:
        return None
    key = (xyz.device.index, xyz.data_ptr(), n)
    hit = _order_cache.pop(key, None)
    if hit is None:
        lib = L.load()
        order = torch.empty(n, dtype=torch.int32, device=xyz.device)
        inv = torch.empty(n, dtype=torch.int32, device=xyz.device)
        ws = torch.empty(lib.a3d_click_spatial_order_workspace_bytes(n), dtype=torch.uint8, device=xyz.device)
        L.check(lib.a3d_click_spatial_order(xyz.data_ptr(), n, order.data_ptr(), inv.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _stream(xyz)), "a3d_click_spatial_order")
        hit = (order, inv)
        while len(_order_cache) >= 256:
            _order_cache.pop(next(iter(_order_cache)))
    _order_cache[key] = hit
    return hit


def _launch_clusters_batch(preds, labels, coords):
    """ONE a3d_click_clusters_batch call (a dozen launches whatever the number of samples) for the non-empty samples, on the
    current stream, and the copy of all their records into one pinned buffer.  Returns (host buffer, slot of every
    sample or None, references that must outlive the stream's work)."""
    dev = preds[0].device
    lib = L.load()
    keep, slots, live = [], [], []
    for i, (pr, lb, xyz) in enumerate(zip(preds, labels, coords)):
        p, l = _i32(pr), _i32(lb)
        x = xyz.to(torch.float32).contiguous()
        n = p.numel()
        if n == 0:
            slots.append(None)
            continue
        if x.shape != (n, 3) or l.numel() != n:
            raise RuntimeError("error_clusters: pred [N], labels [N], coords [N,3] expected")
        slots.append(len(live))
        live.append((p, l, x, _cluster_buffers(dev, n, slot=1 + i)[0]))
    if not live:
        return None, slots, keep
    if len(live) > _MAX_CLICK_BATCH:
        raise RuntimeError(f"error_clusters_batch: {len(live)} samples > {_MAX_CLICK_BATCH} per call")
    key = (dev.index, "cluster_out", len(live))
    buf = _ws_cache.get(key)
    if buf is None:
        buf = _ws_cache[key] = (torch.empty(len(live), _OUT_BYTES, dtype=torch.uint8, device=dev),
                                torch.empty(len(live), _OUT_BYTES, dtype=torch.uint8).pin_memory())
    out, host = buf
    arr = (L.ClickSample * len(live))()
    rec = MAX_CLUSTERS * C.sizeof(L.ClickCluster)
    for k, (p, l, x, work) in enumerate(live):
        sp = arr[k]
        sp.xyz_dev, sp.pred_dev, sp.labels_dev, sp.n = x.data_ptr(), p.data_ptr(), l.data_ptr(), p.numel()
        sp.out_dev, sp.n_out_dev, sp.max_out = out[k].data_ptr(), out[k].data_ptr() + rec, MAX_CLUSTERS
        sp.workspace_dev, sp.workspace_bytes = work.data_ptr(), work.numel()
        so = _spatial_order(x)
        sp.order_dev, sp.inv_dev = (so[0].data_ptr(), so[1].data_ptr()) if so is not None else (None, None)
    L.check(lib.a3d_click_clusters_batch(C.cast(arr, C.c_void_p), len(live), _stream(live[0][0])), "a3d_click_clusters_batch")
    host.copy_(out, non_blocking=True)
    keep.append(live)
    return host, slots, keep


def error_clusters_batch(preds, labels, coords):
    """``error_clusters`` of several samples with ONE set of launches and ONE host synchronisation (round 6: the kernels take
    the samples from a device table -- csrc/clicks.hip; before, every sample ran its dozen launches on a side stream).  Same
    results as the per-sample call, in sample order."""
    if not preds:
        return []
    dev = preds[0].device
    cur = torch.cuda.current_stream(dev)
    try:
        host, slots, keep = _launch_clusters_batch(preds, labels, coords)
    finally:
        cur.synchronize()                        # the inputs and the cached work buffers are in use until here
    hn = host.numpy() if host is not None else None
    return [[] if k is None else _parse_clusters(hn[k]) for k in slots]


def _pick_clicks(clusters, coords_qv, num_obj, current_num_clicks, training):
    """The host half of utils/seg.py:173-226 (ranking, the one ``random.shuffle``, the click dictionaries)."""
    if not clusters:
        return None, None, None, None
    by_id = {c["cluster_id"]: c for c in clusters}
    # ranked by error size, largest first; equal sizes keep ascending-id order (stable sort)
    ranked = sorted(by_id, key=lambda cid: by_id[cid]["error_size"], reverse=True)
    if training:
        chosen = ranked[:num_obj] if len(ranked) >= num_obj else ranked
    else:
        chosen = ranked if current_num_clicks == 0 else ranked[:1]
    random.shuffle(chosen)
    new_clicks, new_pos, new_time = {}, {}, {}
    for order, cid in enumerate(chosen):
        c = by_id[cid]
        key = str(c["label"])
        new_clicks.setdefault(key, []).append(c["row"])
        new_pos.setdefault(key, []).append(coords_qv[c["row"]])
        new_time.setdefault(key, []).append(order)
    return new_clicks, len(chosen), new_pos, new_time


def get_simulated_clicks_batch(preds, labels, coords, current_num_clicks=None, training=True, num_objs=None):
    """``get_simulated_clicks`` for every sample of a batch (engine.py:103-116 / eval_multi_obj.py:162-166 loop over the
    samples): the error clusters of all samples are computed side by side (``error_clusters_batch``), the clicks are then
    picked in sample order, so the global ``random`` stream is consumed exactly as by the per-sample loop.  ``num_objs``
    (training): the number of objects per sample when the caller knows it -- saves a ``torch.unique`` + host round trip
    per sample and round."""
    clusters = error_clusters_batch(preds, labels, coords)
    out = []
    for i, cl in enumerate(clusters):
        num_obj = None
        if training and cl:
            num_obj = num_objs[i] if num_objs is not None else int((torch.unique(labels[i]) != 0).sum())
        out.append(_pick_clicks(cl, coords[i], num_obj, current_num_clicks, training))
    return out


def mean_iou_and_clusters_batch(preds, labels_iou, inverse_maps, labels_qv, coords, n_ids: int = 256):
    """What one round of the interactive protocol needs from the device, with ONE host synchronisation: the error clusters
    of every sample (``error_clusters``: one batched set of launches) and its IoU counts (``mean_iou_scene``), all on the
    caller's stream, their records back through pinned buffers, everything awaited once.  Returns ``(ious, clusters)`` =
    what ``mean_iou_scene_batch`` and ``error_clusters_batch`` return."""
    lib = L.load()
    if not preds:
        return [], []
    dev = preds[0].device
    cur = torch.cuda.current_stream(dev)
    ns = len(preds)
    key = (dev.index, "iou", ns, n_ids)
    buf = _ws_cache.get(key)
    if buf is None:
        buf = _ws_cache[key] = (torch.empty(ns, 3 * n_ids + 1, dtype=torch.int64, device=dev),
                                torch.empty(ns, 3 * n_ids + 1, dtype=torch.int64).pin_memory())
    counts, counts_host = buf
    keep = []
    try:                     # the stream is drained before ANY exit: the cached work buffers are reused by the next call
        host, slots, refs = _launch_clusters_batch(preds, labels_qv, coords)
        keep.append(refs)
        keep.append(_launch_iou_counts(preds, labels_iou, inverse_maps, n_ids, counts))
        counts_host.copy_(counts, non_blocking=True)
    finally:
        cur.synchronize()
    hc = counts_host.numpy()
    if hc[:, -1].any():
        raise RuntimeError("iou_counts: inverse_map holds rows outside the prediction")
    ious = [_mean_iou_from_counts(hc[i, :-1].reshape(3, n_ids).copy()) for i in range(ns)]
    hn = host.numpy() if host is not None else None
    clusters = [[] if k is None else _parse_clusters(hn[k]) for k in slots]
    return ious, clusters


def pick_clicks_batch(clusters, labels, coords, current_num_clicks=None, training=True, num_objs=None):
    """The host half of ``get_simulated_clicks_batch`` for clusters that are already there (sample order: the global
    ``random`` stream is consumed exactly as by the per-sample loop)."""
    out = []
    for i, cl in enumerate(clusters):
        num_obj = None
        if training and cl:
            num_obj = num_objs[i] if num_objs is not None else int((torch.unique(labels[i]) != 0).sum())
        out.append(_pick_clicks(cl, coords[i], num_obj, current_num_clicks, training))
    return out


def get_simulated_clicks(pred_qv, labels_qv, coords_qv, current_num_clicks=None, training=True):
    """utils/seg.py:173-226.  Same returns: (new_clicks {str(label): [rows]}, click_num,
    new_click_pos {str(label): [xyz tensors]}, new_click_time {str(label): [order]}), or four Nones when
    the prediction is already right.  Consumes the global ``random`` stream like the reference
    (one ``random.shuffle`` of the selected cluster ids)."""
    clusters = error_clusters(pred_qv, labels_qv, coords_qv)
    num_obj = int((torch.unique(labels_qv) != 0).sum()) if training and clusters else None
    return _pick_clicks(clusters, coords_qv, num_obj, current_num_clicks, training)


def extend_clicks(current_clicks, current_clicks_time, new_clicks, new_click_time):
    """utils/seg.py:229-239."""
    offset = sum(len(v) for v in current_clicks_time.values())
    for obj_id, rows in new_clicks.items():
        current_clicks[obj_id].extend(rows)
        current_clicks_time[obj_id].extend(t + offset for t in new_click_time[obj_id])
    return current_clicks, current_clicks_time


def cal_click_loss_weights(batch_idx, raw_coords, labels, click_idx, alpha=0.8, beta=2.0, tita=0.3, ranges=None):
    """utils/seg.py:72-89: per sample, weights[i] = alpha + (beta-alpha)(1 - min(d_i, tita)/tita) with
    d_i the distance of point i to the nearest click of any object.  ``ranges`` [(first row, end row)] per sample, when the
    caller knows them (the rows of a sample are contiguous in a collated batch): no boolean-mask gathers, no host syncs."""
    lib = L.load()
    if not raw_coords.is_cuda:
        raise RuntimeError("cal_click_loss_weights runs on the GPU only")
    weights = []
    for i in range(len(ranges) if ranges is not None else int(batch_idx.max()) + 1):
        xyz = (raw_coords[ranges[i][0]:ranges[i][1]] if ranges is not None else raw_coords[batch_idx == i]).to(torch.float32).contiguous()
        rows = [int(r) for v in click_idx[i].values() for r in v]
        r, rp = _host_i32(rows)
        w = torch.empty(xyz.shape[0], dtype=torch.float32, device=xyz.device)
        L.check(lib.a3d_click_loss_weights(xyz.data_ptr(), xyz.shape[0], rp, len(rows), tita, alpha, beta,
                                           w.data_ptr(), _stream(xyz)), "a3d_click_loss_weights")
        weights.append(w)
    return weights
