"""Seeded synthetic scenes for benchmarks and parity tests.

The reference ships no synthetic data; BASELINE.json's metric is quoted on a
"synthetic 80k-voxel scene, 10 click queries".  SURVEY.md section 8(d) defines the
generator: a hollow axis-aligned room shell plus 25 hollow boxes ("furniture")
on an integer voxel grid, scaled until the number of occupied voxels is within
1 % of the target; rgb ~ U[0,1)^3; raw xyz = voxel_size * grid + U(0, voxel_size).

What the model consumes is what ``datasets/InterMultiObj3DSegDataset.py:49-75,
126-136`` of the reference would hand it: unique int32 voxel coordinates with a
leading batch column, one fp32 colour triple per voxel, and the raw metric
coordinates of the representative point of each voxel.
"""
from __future__ import annotations

import numpy as np


def _shell(lo, hi):
    """Integer coordinates of the surface of the axis aligned box [lo, hi] (inclusive)."""
    lo = np.asarray(lo, dtype=np.int64)
    hi = np.asarray(hi, dtype=np.int64)
    faces = []
    for ax in range(3):
        o = [a for a in range(3) if a != ax]
        g0 = np.arange(lo[o[0]], hi[o[0]] + 1)
        g1 = np.arange(lo[o[1]], hi[o[1]] + 1)
        a, b = np.meshgrid(g0, g1, indexing="ij")
        for v in (lo[ax], hi[ax]):
            pts = np.empty((a.size, 3), dtype=np.int64)
            pts[:, ax] = v
            pts[:, o[0]] = a.ravel()
            pts[:, o[1]] = b.ravel()
            faces.append(pts)
    return np.concatenate(faces, 0)


def _scene_at_scale(rng_seed: int, scale: float, n_boxes: int = 25):
    rng = np.random.default_rng(rng_seed)
    room = np.array([8.0, 6.0, 2.6]) * scale  # a room of 8 x 6 x 2.6 "units"
    room_i = np.maximum(np.round(room).astype(np.int64), 4)
    parts = [_shell([0, 0, 0], room_i)]
    owner = [np.zeros(len(parts[0]), dtype=np.int32)]
    for b in range(n_boxes):
        size = rng.uniform([0.4, 0.4, 0.3], [2.0, 1.6, 1.4]) * scale
        size_i = np.maximum(np.round(size).astype(np.int64), 1)
        hi_lim = np.maximum(room_i - size_i - 1, 2)
        lo = np.array([rng.integers(1, hi_lim[0]), rng.integers(1, hi_lim[1]), 1])
        if rng.random() < 0.3:  # some objects hang off the floor
            lo[2] = rng.integers(1, max(2, hi_lim[2]))
        pts = _shell(lo, lo + size_i)
        parts.append(pts)
        owner.append(np.full(len(pts), b + 1, dtype=np.int32))
    pts = np.concatenate(parts, 0)
    own = np.concatenate(owner, 0)
    # unique voxels; a voxel keeps the label of the last box that touched it
    key = (pts[:, 0] << 42) | (pts[:, 1] << 21) | pts[:, 2]
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    last = np.ones(len(key_s), dtype=bool)
    last[:-1] = key_s[1:] != key_s[:-1]
    sel = order[last]
    return pts[sel].astype(np.int32), own[sel]


def make_scene(n_target: int = 80_000, seed: int = 0, voxel_size: float = 0.02,
               n_boxes: int = 25, batch_index: int = 0, shuffle: bool = True):
    """Return a dict with the tensors one scene of the hot path consumes.

    coords  int32 [N,4]  (batch, x, y, z)   -- unique voxels
    feats   fp32  [N,3]  rgb in [0,1)
    raw_xyz fp32  [N,3]  metres, min-shifted like the reference dataset does
    labels  int32 [N]    0 = room shell, 1..n_boxes = furniture id (for click simulation)
    """
    # voxel count grows ~ scale^2 (surfaces): secant steps on that law, then bisection if needed
    best, scale, lo, hi = None, 40.0, None, None
    for it in range(40):
        c, own = _scene_at_scale(seed, scale, n_boxes)
        if best is None or abs(len(c) - n_target) < abs(len(best[0]) - n_target):
            best = (c, own)
        if abs(len(c) - n_target) <= 0.01 * n_target:
            break
        if len(c) < n_target:
            lo = scale if lo is None else max(lo, scale)
        else:
            hi = scale if hi is None else min(hi, scale)
        guess = scale * (n_target / max(len(c), 1)) ** 0.5
        if lo is not None and hi is not None:
            if not (lo < guess < hi) or it > 6:
                guess = 0.5 * (lo + hi)
        scale = min(max(guess, 0.5), 2000.0)
    c, own = best
    rng = np.random.default_rng(seed + 1_000_003)
    if shuffle:  # dataset order is not spatially sorted; do not let tests rely on it
        p = rng.permutation(len(c))
        c, own = c[p], own[p]
    n = len(c)
    coords = np.empty((n, 4), dtype=np.int32)
    coords[:, 0] = batch_index
    coords[:, 1:] = c
    feats = rng.random((n, 3), dtype=np.float32)
    raw = (c.astype(np.float32) + rng.random((n, 3), dtype=np.float32)) * np.float32(voxel_size)
    raw = raw - raw.min(0, keepdims=True)
    return {"coords": coords, "feats": feats, "raw_xyz": raw.astype(np.float32), "labels": own}


def make_clicks(labels: np.ndarray, n_objects: int = 5, clicks_per_object: int = 2,
                n_bg_clicks: int = 0, seed: int = 0):
    """Click dictionaries in the reference's format (``eval_multi_obj.py:105-109``):
    click_idx  {'0': [bg rows], '1': [rows of object 1], ...}; click_time_idx the same
    shape holding the global click order."""
    rng = np.random.default_rng(seed + 77)
    ids = [i for i in np.unique(labels) if i > 0 and (labels == i).sum() >= clicks_per_object]
    ids = list(rng.permutation(ids)[:n_objects])
    assert len(ids) == n_objects, "scene has too few objects"
    click_idx = {"0": []}
    click_time = {"0": []}
    t = 0
    for k, oid in enumerate(ids, start=1):
        rows = np.flatnonzero(labels == oid)
        pick = rng.choice(rows, size=clicks_per_object, replace=False)
        click_idx[str(k)] = [int(r) for r in pick]
        click_time[str(k)] = list(range(t, t + clicks_per_object))
        t += clicks_per_object
    if n_bg_clicks:
        rows = np.flatnonzero(~np.isin(labels, ids))
        pick = rng.choice(rows, size=n_bg_clicks, replace=False)
        click_idx["0"] = [int(r) for r in pick]
        click_time["0"] = list(range(t, t + n_bg_clicks))
    return click_idx, click_time
