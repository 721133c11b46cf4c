"""The load-time ``kernel_order`` switch ("Hard parts"): known-answer cases with two voxels.

The library and the oracle enumerate kernel offsets with x fastest: k = ix + s*iy + s^2*iz, offset = (ix, iy, iz) - s//2
(odd kernels) or (ix, iy, iz) (2^3 kernels).  Whether MinkowskiEngine wrote the authors' checkpoint in that enumeration
can only be decided with weights + data; a checkpoint in the other (z fastest) enumeration is re-indexed while loading."""
import numpy as np
import torch

from agile3d_amd import build_model, default_args
from agile3d_amd.model import convert_kernel_order, kernel_order_permutation
from oracle import backbone as ob


def _two_voxels(offset):
    coords = np.array([[0, 4, 4, 4], [0, 4 + offset[0], 4 + offset[1], 4 + offset[2]]], np.int32)
    lv = ob.SparseLevels(coords, n_levels=1)
    rows = {tuple(c[1:]): i for i, c in enumerate(lv.levels[0])}
    return lv, rows[(4, 4, 4)], rows[(4 + offset[0], 4 + offset[1], 4 + offset[2])]


def test_permutation_is_an_involution_and_fixes_the_centre():
    for vol in (8, 27, 125):
        p = kernel_order_permutation(vol)
        assert sorted(p) == list(range(vol))
        assert [p[p[k]] for k in range(vol)] == list(range(vol))
    assert kernel_order_permutation(27)[13] == 13 and kernel_order_permutation(1) == [0]


def test_two_voxel_known_answer_3x3x3():
    """Voxel B sits at A + (1, 0, 0).  With the one-hot kernel of offset (+1, 0, 0) the output at A is B's feature.
    x fastest stores that offset at k = 2 + 3*1 + 9*1 = 14, z fastest at k' = 1 + 3*1 + 9*2 = 22."""
    lv, a, b = _two_voxels((1, 0, 0))
    x = torch.zeros(2, 4)
    x[a], x[b] = torch.tensor([1., 2., 3., 4.]), torch.tensor([5., 6., 7., 8.])
    kmap = lv.kernel_map(0, 3)
    w_x = torch.zeros(27, 4, 4)
    w_x[14] = torch.eye(4)
    out = ob.sparse_conv(x, w_x, kmap, 2)
    assert torch.equal(out[a], x[b]) and torch.equal(out[b], torch.zeros(4))
    w_z = torch.zeros(27, 4, 4)                      # the same layer as a z-fastest file would hold it
    w_z[22] = torch.eye(4)
    assert not torch.equal(ob.sparse_conv(x, w_z, kmap, 2)[a], x[b])          # loaded as is: wrong neighbour
    conv = convert_kernel_order({"backbone.block1.0.conv1.kernel": w_z, "backbone.bn0.bn.weight": torch.ones(4),
                                 "lin_squeeze_head.kernel": torch.ones(4, 4)}, "z_fastest")
    assert torch.equal(conv["backbone.block1.0.conv1.kernel"], w_x)
    assert torch.equal(conv["lin_squeeze_head.kernel"], torch.ones(4, 4))     # 1x1 kernels and other tensors untouched
    assert torch.equal(ob.sparse_conv(x, conv["backbone.block1.0.conv1.kernel"], kmap, 2)[a], x[b])
    same = convert_kernel_order({"k.kernel": w_x}, "x_fastest")
    assert torch.equal(same["k.kernel"], w_x)


def test_known_answer_5x5x5_and_2x2x2():
    p125, p8 = kernel_order_permutation(125), kernel_order_permutation(8)
    # 5^3: offset (+2, -1, 0) -> (ix, iy, iz) = (4, 1, 2): x fastest 4 + 5 + 50 = 59, z fastest 2 + 5 + 100 = 107
    assert p125[59] == 107
    # 2^3 (stride 2, SURVEY App. B.4: child slot k = x + 2y + 4z): slot of child (1, 0, 0) is 1, z fastest it is 4
    assert p8[1] == 4 and p8[2] == 2 and p8[4] == 1 and p8[0] == 0 and p8[7] == 7
    lv, a, b = _two_voxels((2, -1, 0))
    x = torch.zeros(2, 3)
    x[b] = torch.tensor([1., -2., 3.])
    w_z = torch.zeros(125, 3, 3)
    w_z[107] = torch.eye(3)
    w = convert_kernel_order({"backbone.conv0p1s1.kernel": w_z}, "z_fastest")["backbone.conv0p1s1.kernel"]
    assert torch.equal(ob.sparse_conv(x, w, lv.kernel_map(0, 5), 2)[a], x[b])


def test_model_load_state_dict_switch(full_model_cpu):
    """Agile3d.load_state_dict(kernel_order='z_fastest') re-indexes every sparse-conv kernel of a full state dict."""
    from agile3d_amd import build_model, default_args
    sd = {k: v.clone() for k, v in full_model_cpu.state_dict().items()}
    z_file = convert_kernel_order(sd, "z_fastest")           # the permutation is an involution: this IS the z-fastest file
    m = build_model(default_args())
    m.load_state_dict(z_file, strict=True, kernel_order="z_fastest")
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    changed = [k for k in sd if not torch.equal(sd[k], z_file[k])]
    assert len(changed) == 46 + 1 + 8 and all(k.endswith(".kernel") for k in changed)     # 3^3, 5^3 and the 2^3 kernels


def test_load_state_dict_does_not_reapply_a_sticky_order(monkeypatch):
    """A state dict written by this library is in its own order: neither args.kernel_order nor A3D_KERNEL_ORDER may permute
    it on load (resume after importing the authors' weights, model_b.load_state_dict(model_a.state_dict())); the permutation
    happens only on an explicit argument or through import_state_dict."""
    torch.manual_seed(3)
    args = default_args()
    args.kernel_order = "z_fastest"
    monkeypatch.setenv("A3D_KERNEL_ORDER", "z_fastest")
    a, b = build_model(args), build_model(args)
    sd = {k: v.clone() for k, v in a.state_dict().items()}
    b.load_state_dict(sd, strict=True)
    name = "backbone.block1.0.conv1.kernel"
    assert torch.equal(b.state_dict()[name], sd[name])
    b.import_state_dict(sd, strict=True)                      # the explicit import path reads the sticky setting
    assert torch.equal(b.state_dict()[name], sd[name][kernel_order_permutation(27)])
    assert not torch.equal(b.state_dict()[name], sd[name])
