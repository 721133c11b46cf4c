"""CPU oracle of ME.utils.sparse_quantize as the reference's dataset uses it
(datasets/InterMultiObj3DSegDataset.py:67-75; semantics SURVEY.md App. B.2).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against MinkowskiEngine itself (third-party, not in /root/reference, not installable here):
what is restated is the documented contract the dataset code relies on -- floor division in the input's dtype,
int32 voxels, ``unique_map`` = a representative point per voxel, ``inverse_map`` = voxel of every point -- with
the representative fixed as the FIRST point of the voxel and voxels numbered by first occurrence.
Plain Python on purpose (a dict in input order), independent of the numpy/rocPRIM implementations it checks.
"""
import numpy as np


def sparse_quantize(coords: np.ndarray, quantization_size: float):
    q = np.floor(coords / quantization_size).astype(np.int32)      # numpy keeps float32 / python float in float32
    seen = {}
    unique_map, inverse_map = [], []
    for i, row in enumerate(map(tuple, q.tolist())):
        v = seen.get(row)
        if v is None:
            v = seen[row] = len(unique_map)
            unique_map.append(i)
        inverse_map.append(v)
    unique_map = np.asarray(unique_map, dtype=np.int64)
    return q[unique_map], unique_map, np.asarray(inverse_map, dtype=np.int64)
