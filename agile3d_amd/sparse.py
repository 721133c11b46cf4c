"""The slice of MinkowskiEngine's Python surface that callers of the hot path touch.

Reference call sites: ``engine.py:47-51``, ``eval_multi_obj.py:94-98`` build
``ME.SparseTensor(coordinates=, features=, device=)``; datasets call
``ME.utils.sparse_quantize(coordinates, quantization_size, return_index=True,
return_inverse=True)`` and ``ME.utils.batched_coordinates``
(``datasets/InterMultiObj3DSegDataset.py:67-71,129``).  Semantics: SURVEY.md App. B.1-2.
"""
from __future__ import annotations

import numpy as np
import torch


class SparseTensor:
    """Coordinates int32 [N,4] (batch, x, y, z) + features fp32 [N,C], rows in caller order.

    Exposes ``.C``, ``.F``, ``.device``, ``.decomposed_features`` (the attributes the reference
    reads: ``agile3d.py:146-150,185,194-199``).  Voxel coordinates must be unique
    (as ``sparse_quantize`` produces); the HIP scene builder reports duplicates as an error.
    """

    def __init__(self, features=None, coordinates=None, device=None, **_ignored):
        if coordinates is None or features is None:
            raise ValueError("SparseTensor needs coordinates and features")
        if not torch.is_tensor(coordinates):
            coordinates = torch.as_tensor(np.asarray(coordinates))
        if not torch.is_tensor(features):
            features = torch.as_tensor(np.asarray(features))
        if device is not None:
            coordinates = coordinates.to(device)
            features = features.to(device)
        if coordinates.dim() != 2 or coordinates.shape[1] != 4:
            raise ValueError("coordinates must be [N,4] (batch index first; use batched_coordinates)")
        if features.shape[0] != coordinates.shape[0]:
            raise ValueError("coordinates / features row count mismatch")
        self.C = coordinates.to(torch.int32).contiguous()
        self.F = features.to(torch.float32).contiguous()
        self._extra = {}

    @property
    def device(self):
        return self.F.device

    @property
    def coordinates(self):
        return self.C

    @property
    def features(self):
        return self.F

    def batch_ranges(self):
        """[(start, end)] row range of every batch sample (rows of one sample are contiguous,
        as ``batched_coordinates`` produces)."""
        if "ranges" not in self._extra:
            b = self.C[:, 0]
            if b.numel() == 0:
                self._extra["ranges"] = []
            else:
                # one device->host copy: per-sample counts followed by the number of batch-index
                # changes (rows of one sample must be contiguous, as batched_coordinates produces)
                counts = torch.bincount(b.to(torch.int64))
                chg = (b[1:] != b[:-1]).sum().reshape(1)
                host = torch.cat([counts, chg]).cpu().tolist()
                counts, chg = host[:-1], host[-1]
                if chg != len(counts) - 1 or min(counts) == 0:
                    raise ValueError("rows of each batch sample must be contiguous and batch indices dense")
                r, s = [], 0
                for c in counts:
                    r.append((s, s + c))
                    s += c
                self._extra["ranges"] = r
        return self._extra["ranges"]

    @property
    def decomposed_features(self):
        return [self.F[s:e] for (s, e) in self.batch_ranges()]

    def __len__(self):
        return self.C.shape[0]


def sparse_quantize(coordinates, features=None, labels=None, quantization_size=None,
                    return_index=False, return_inverse=False, **_ignored):
    """``ME.utils.sparse_quantize``: ``floor(coords / quantization_size)`` in the
    input's own float dtype, int32 cast, unique voxels; ``unique_map`` = index of the first point
    of each voxel (in input order), ``inverse_map`` = voxel row of every point.

    A CUDA tensor is voxelised on the GPU (``a3d_sparse_quantize``, csrc/quantize.hip; results stay on
    the device); numpy arrays / CPU tensors take the host path the reference's dataset code runs in its
    loader workers.  Both produce identical integers."""
    if torch.is_tensor(coordinates) and coordinates.is_cuda:
        return _sparse_quantize_device(coordinates, features, labels, quantization_size, return_index, return_inverse)
    c = np.asarray(coordinates.cpu() if torch.is_tensor(coordinates) else coordinates)
    if quantization_size is not None:
        c = np.floor(c / quantization_size)
    q = c.astype(np.int32)
    # first-occurrence unique, output ordered by first occurrence
    _, first_idx, inverse = np.unique(q, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first_idx, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    unique_map = first_idx[order]
    inverse_map = rank[inverse.reshape(-1)]
    uq = q[unique_map]
    out = [torch.from_numpy(uq) if torch.is_tensor(coordinates) else uq]
    if features is not None:
        out.append(features[unique_map])
    if labels is not None:
        out.append(labels[unique_map])
    if return_index:
        out.append(torch.from_numpy(unique_map) if torch.is_tensor(coordinates) else unique_map)
    if return_inverse:
        out.append(torch.from_numpy(inverse_map) if torch.is_tensor(coordinates) else inverse_map)
    return out[0] if len(out) == 1 else tuple(out)


def _sparse_quantize_device(coordinates, features, labels, quantization_size, return_index, return_inverse):
    import ctypes as C

    from . import lib as L
    lib = L.load()
    xyz = coordinates
    if xyz.dim() != 2 or xyz.shape[1] != 3 or xyz.dtype not in (torch.float32, torch.float64):
        raise ValueError("sparse_quantize on the GPU takes float32/float64 [N,3] coordinates")
    xyz = xyz.contiguous()
    n = xyz.shape[0]
    dev = xyz.device
    if n == 0:
        raise ValueError("sparse_quantize: empty point cloud")
    ws = torch.empty(lib.a3d_quantize_workspace_bytes(n), dtype=torch.uint8, device=dev)
    q = torch.empty((n, 3), dtype=torch.int32, device=dev)
    umap = torch.empty(n, dtype=torch.int64, device=dev)
    inv = torch.empty(n, dtype=torch.int64, device=dev)
    nv = C.c_int64()
    size = 1.0 if quantization_size is None else float(quantization_size)
    L.check(lib.a3d_sparse_quantize(xyz.data_ptr(), int(xyz.dtype == torch.float64), n, size, q.data_ptr(),
                                    umap.data_ptr(), inv.data_ptr(), C.byref(nv), ws.data_ptr(), ws.numel(),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "a3d_sparse_quantize")
    m = nv.value
    unique_map = umap[:m]
    out = [q[:m]]
    if features is not None:
        out.append(features[unique_map])
    if labels is not None:
        out.append(labels[unique_map])
    if return_index:
        out.append(unique_map)
    if return_inverse:
        out.append(inv)
    return out[0] if len(out) == 1 else tuple(out)


def batched_coordinates(coords, dtype=torch.int32, device=None):
    """``ME.utils.batched_coordinates``: list of [n_i,3] -> int32 [sum n_i, 4] with the batch
    index prepended."""
    parts = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(np.asarray(c) if not torch.is_tensor(c) else c).to(dtype)
        col = torch.full((c.shape[0], 1), b, dtype=dtype)
        parts.append(torch.cat([col, c], 1))
    out = torch.cat(parts, 0) if parts else torch.zeros((0, 4), dtype=dtype)
    return out.to(device) if device is not None else out
