"""Independent pins for the sparse-backbone oracle (MinkowskiEngine is not installable here):
dense conv3d / conv_transpose3d equivalence on densified scenes and
hand-computable known-answer cases."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import backbone as ob


def random_scene(n, extent, seed, batch=1):
    rng = np.random.default_rng(seed)
    pts = set()
    while len(pts) < n:
        p = tuple(rng.integers(0, extent, 3).tolist())
        pts.add(p)
    c = np.array(sorted(pts), dtype=np.int32)
    c = c[rng.permutation(len(c))]
    return np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)


def densify(coords, feats, extent):
    C = feats.shape[1]
    d = torch.zeros(1, C, extent, extent, extent)
    d[0, :, coords[:, 3], coords[:, 2], coords[:, 1]] = feats.T
    return d


@pytest.mark.parametrize("ksize,cin,cout", [(3, 5, 7), (5, 3, 4)])
def test_odd_kernel_equals_dense_conv3d(ksize, cin, cout):
    E = 12
    coords = random_scene(300, E, 0)
    lv = ob.SparseLevels(coords, n_levels=2)
    x = torch.randn(len(coords), cin)
    W = torch.randn(ksize ** 3, cin, cout)
    y = ob.sparse_conv(x, W, lv.kernel_map(0, ksize), lv.n(0))
    wd = W.view(ksize, ksize, ksize, cin, cout).permute(4, 3, 0, 1, 2)   # [co,ci,kz,ky,kx]
    yd = F.conv3d(densify(coords, x, E), wd, padding=ksize // 2)
    ref = yd[0][:, coords[:, 3], coords[:, 2], coords[:, 1]].T
    assert torch.allclose(y, ref, atol=1e-4), (y - ref).abs().max()


def test_stride2_and_transpose_equal_dense():
    E = 16
    coords = random_scene(400, E, 1)
    lv = ob.SparseLevels(coords, n_levels=2)
    cin, cout = 6, 5
    x = torch.randn(len(coords), cin)
    W = torch.randn(8, cin, cout)
    y = ob.sparse_conv(x, W, lv.stride_map(0), lv.n(1))
    wd = W.view(2, 2, 2, cin, cout).permute(4, 3, 0, 1, 2)
    yd = F.conv3d(densify(coords, x, E), wd, stride=2)
    cc = lv.levels[1]
    ref = yd[0][:, cc[:, 3], cc[:, 2], cc[:, 1]].T
    assert torch.allclose(y, ref, atol=1e-4)
    # every coarse voxel has >= 1 child and only those exist
    occ = F.max_pool3d((densify(coords, torch.ones(len(coords), 1), E)), 2)
    assert int(occ.sum()) == lv.n(1)
    # transposed conv back onto the fine coordinate set
    Wt = torch.randn(8, cout, cin)
    kmap = [(rc, rf) for (rf, rc) in lv.stride_map(0)]
    z = ob.sparse_conv(y, Wt, kmap, lv.n(0))
    wtd = Wt.view(2, 2, 2, cout, cin).permute(3, 4, 0, 1, 2)              # [ci,co,kz,ky,kx]
    zd = F.conv_transpose3d(densify(cc, y, E // 2), wtd, stride=2)
    refz = zd[0][:, coords[:, 3], coords[:, 2], coords[:, 1]].T
    assert torch.allclose(z, refz, atol=1e-4)


def test_known_answer_offset_indexing():
    """Two voxels at (0,0,0) and (1,0,0); one-hot kernels pin k <-> (dx,dy,dz), x fastest."""
    coords = np.array([[0, 0, 0, 0], [0, 1, 0, 0]], np.int32)
    lv = ob.SparseLevels(coords, n_levels=1)
    x = torch.tensor([[1.0], [10.0]])
    for k, expect in {13: [1.0, 10.0], 14: [10.0, 0.0], 12: [0.0, 1.0], 16: [0.0, 0.0]}.items():
        W = torch.zeros(27, 1, 1)
        W[k] = 1.0
        y = ob.sparse_conv(x, W, lv.kernel_map(0, 3), 2)
        assert y.reshape(-1).tolist() == expect, k


def test_single_voxel_and_block_roundtrip():
    c = np.array([[0, 4, 6, 2]], np.int32)
    lv = ob.SparseLevels(c, n_levels=3)
    assert [lv.n(i) for i in range(3)] == [1, 1, 1]
    assert lv.levels[1].tolist() == [[0, 2, 3, 1]] and lv.levels[2].tolist() == [[0, 1, 1, 0]]
    # a full 2x2x2 block collapses to one coarse voxel; down(W=1) sums the 8 children
    blk = np.array([[0, x, y, z] for z in (0, 1) for y in (0, 1) for x in (0, 1)], np.int32)
    lv = ob.SparseLevels(blk, n_levels=2)
    x = torch.arange(8.0).reshape(8, 1)
    y = ob.sparse_conv(x, torch.ones(8, 1, 1), lv.stride_map(0), 1)
    assert float(y) == 28.0
    # slot index k = x + 2y + 4z: W[k]=k picks sum k*x_k = sum k^2
    y = ob.sparse_conv(x, torch.arange(8.0).reshape(8, 1, 1), lv.stride_map(0), 1)
    assert float(y) == float((np.arange(8) ** 2).sum())


def test_negative_coordinates_floor():
    c = np.array([[0, -1, -2, -3], [0, 0, 0, 0]], np.int32)
    lv = ob.SparseLevels(c, n_levels=2)
    assert sorted(lv.levels[1].tolist()) == [[0, -1, -1, -2], [0, 0, 0, 0]]


def test_batch_norm_eval_matches_torch(full_model_cpu):
    sd = full_model_cpu.state_dict()
    x = torch.randn(50, 32)
    y = ob.batch_norm_eval(x, sd, "backbone.bn0.")
    bn = full_model_cpu.backbone.bn0.bn
    assert torch.allclose(y, bn(x), atol=1e-6)


def test_full_backbone_small_scene_runs_and_matches_dense_first_layers(full_model_cpu):
    """Whole Res16UNet34C on a 600-voxel scene: shapes, finiteness, and the stem + first down
    conv re-derived densely."""
    sd = full_model_cpu.state_dict()
    E = 24
    coords = random_scene(600, E, 3)
    feats = torch.rand(len(coords), 3)
    raw = torch.from_numpy(coords[:, 1:].astype(np.float32)) * 0.05
    r = ob.forward_backbone(sd, coords, feats, raw)
    assert r["pcd_features"].shape == (600, 128) and r["pos_enc"].shape == (600, 128)
    assert [f.shape[1] for f in r["feature_maps"]] == [256, 256, 128, 96, 96]
    assert torch.isfinite(r["pcd_features"]).all()
    W0 = sd["backbone.conv0p1s1.kernel"]
    wd = W0.view(5, 5, 5, 3, 32).permute(4, 3, 0, 1, 2)
    yd = F.conv3d(densify(coords, feats, E), wd, padding=2)
    stem = yd[0][:, coords[:, 3], coords[:, 2], coords[:, 1]].T
    stem = torch.relu(full_model_cpu.backbone.bn0.bn(stem))
    lv = r["levels"]
    mine = torch.relu(ob.batch_norm_eval(ob.sparse_conv(feats, W0, lv.kernel_map(0, 5), 600), sd, "backbone.bn0."))
    assert torch.allclose(mine, stem, atol=1e-4)
