"""Voxelisation: host path vs the plain-Python oracle (CPU), device path vs
both (GPU, bit-exact integers)."""
import numpy as np
import pytest
import torch

from agile3d_amd import sparse_quantize
from oracle import quantize as oq


def clouds():
    rng = np.random.default_rng(0)
    yield "f32_room", (rng.random((20000, 3)) * [8, 6, 2.6] - [4, 3, 0]).astype(np.float32), 0.05
    yield "f64_room", rng.random((15000, 3)) * [8, 6, 2.6] - [4, 3, 0], 0.05
    yield "dense_dups", (rng.integers(-20, 20, (30000, 3)) * 0.02 + 0.01).astype(np.float32), 0.02
    yield "on_boundaries", (rng.integers(-50, 50, (5000, 3)) * 0.05).astype(np.float32), 0.05
    yield "one_voxel", np.full((100, 3), 0.3, np.float32), 1.0
    yield "single_point", np.array([[-1.5, 2.25, 0.0]], np.float32), 0.5
    yield "large_extent", (rng.random((4000, 3)) * 4000 - 2000).astype(np.float32), 0.01


@pytest.mark.parametrize("name,xyz,qs", list(clouds()), ids=[c[0] for c in clouds()])
def test_host_path_matches_oracle(name, xyz, qs):
    q, umap, inv = sparse_quantize(xyz, quantization_size=qs, return_index=True, return_inverse=True)
    wq, wu, wi = oq.sparse_quantize(xyz, qs)
    assert q.dtype == np.int32 and np.array_equal(q, wq) and np.array_equal(umap, wu) and np.array_equal(inv, wi)
    assert np.array_equal(q[inv], np.floor(xyz / qs).astype(np.int32))       # every point maps to its own voxel
    assert len(np.unique(q, axis=0)) == len(q)                               # voxels are unique


@pytest.mark.gpu
@pytest.mark.parametrize("name,xyz,qs", list(clouds()), ids=[c[0] for c in clouds()])
def test_device_path_matches_host_and_oracle(name, xyz, qs):
    t = torch.from_numpy(xyz).cuda()
    feats = torch.arange(len(xyz), dtype=torch.float32).cuda()
    q, f, umap, inv = sparse_quantize(t, features=feats, quantization_size=qs, return_index=True, return_inverse=True)
    wq, wu, wi = oq.sparse_quantize(xyz, qs)
    assert q.is_cuda and q.dtype == torch.int32 and umap.dtype == torch.int64 and inv.dtype == torch.int64
    assert np.array_equal(q.cpu().numpy(), wq) and np.array_equal(umap.cpu().numpy(), wu)
    assert np.array_equal(inv.cpu().numpy(), wi) and torch.equal(f, feats[umap])
    hq, hu, hi = sparse_quantize(xyz, quantization_size=qs, return_index=True, return_inverse=True)
    assert np.array_equal(q.cpu().numpy(), hq) and np.array_equal(umap.cpu().numpy(), hu) and np.array_equal(inv.cpu().numpy(), hi)


@pytest.mark.gpu
def test_device_path_full_resolution_scan_and_errors():
    """1.2 M points (a full-resolution ScanNet scan) -> properties that do not need the oracle:
    every point lands in its own voxel, unique_map holds the smallest point index of each voxel and is increasing."""
    rng = np.random.default_rng(1)
    xyz = (rng.random((1_200_000, 3)) * [8, 6, 2.6]).astype(np.float32)
    t = torch.from_numpy(xyz).cuda()
    q, umap, inv = sparse_quantize(t, quantization_size=0.05, return_index=True, return_inverse=True)
    want = torch.from_numpy(np.floor(xyz / 0.05).astype(np.int32)).cuda()   # numpy: true fp32 division (torch on the
    assert torch.equal(q[inv], want)                                        # GPU multiplies by the reciprocal)
    assert bool((umap[1:] > umap[:-1]).all())                      # first-occurrence order
    first = torch.full((len(q),), len(xyz), dtype=torch.int64, device="cuda").scatter_reduce(
        0, inv, torch.arange(len(xyz), device="cuda"), reduce="amin")
    assert torch.equal(first, umap)
    assert len(torch.unique(q, dim=0)) == len(q)
    from agile3d_amd.lib import A3DError
    bad = t.clone()
    bad[7, 1] = float("nan")
    with pytest.raises(A3DError):
        sparse_quantize(bad, quantization_size=0.05, return_index=True)
    with pytest.raises(A3DError):
        sparse_quantize(t * 1e6, quantization_size=0.05)


def test_me_style_utils_namespace():
    """`import agile3d_amd as ME`: the dataset code's ME.utils.sparse_quantize / ME.utils.batched_coordinates calls
    (datasets/InterMultiObj3DSegDataset.py:67-71,129) resolve to the same functions."""
    import agile3d_amd as ME
    pts = np.array([[0.01, 0.02, 0.03], [0.04, 0.01, 0.02], [0.26, 0.0, 0.0]], np.float32)
    q, idx, inv = ME.utils.sparse_quantize(coordinates=pts, quantization_size=0.05, return_index=True, return_inverse=True)
    assert q.tolist() == [[0, 0, 0], [5, 0, 0]] and idx.tolist() == [0, 2] and inv.tolist() == [0, 0, 1]
    bc = ME.utils.batched_coordinates([q, q[:1]])
    assert bc.tolist() == [[0, 0, 0, 0], [0, 5, 0, 0], [1, 0, 0, 0]]
    assert ME.SparseTensor is not None and ME.utils.sparse_quantize is ME.sparse_quantize
