"""Helper of tests/test_gpu_conv.py::test_emulated_fp32_products_error_bound (not a test module).

The library reads A3D_CONV_EMU once per process, so the two builds are compared across interpreters:
``python tests/emu_cases.py out.npz`` runs every gathered-convolution shape class of the backbone (res16unet.py:89-147: the
3^3 convolutions of the five levels, the stride-2 pairs) on ADVERSARIAL inputs and stores the raw outputs; the test runs it
with A3D_CONV_EMU=0 and =2 and measures both against a float64 evaluation of the same sums.

Input families (x = features, w = weights; both fp32):
  wide       every element scaled by its own power of two in 2^-24 .. 2^24: products span 2^-48 .. 2^48 inside one sum
  cancel     channels in pairs (x, -x) against weights (w, w (1 + u 2^-12)): the big products cancel, what is left is 2^-12 of them
  same_sign  |x|, |w|: nothing cancels -- a truncation bias in the operand split would add up over the whole sum
  tiny       x ~ 2^-60, w ~ 2^40: the low bf16 planes of x sit near 2^-76, still normal numbers
  subnormal  x ~ 2^-118, w ~ 2^100: the low planes of x fall below 2^-126 -- OUTSIDE the domain the emulation is specified
             for (|x| >= 2^-100 or 0); reported, not bounded
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

FAMILIES = ["wide", "cancel", "same_sign", "tiny", "subnormal"]
# (kind, level_in, cin, cout): one per (kernel build, role) the backbone program contains
SHAPES = [("conv3", 0, 32, 32), ("conv3", 1, 32, 64), ("conv3", 2, 64, 64), ("conv3", 0, 128, 96), ("conv3", 1, 96, 96),
          ("conv3", 3, 64, 128), ("conv3", 2, 128, 128), ("conv3", 2, 192, 128), ("conv3", 4, 128, 256), ("conv3", 4, 256, 256),
          ("conv3", 3, 384, 256), ("down", 0, 32, 32), ("down", 2, 64, 64), ("down", 3, 128, 128), ("up", 4, 256, 256),
          ("up", 3, 256, 128), ("up", 2, 128, 96), ("up", 1, 96, 96)]


def name_of(shape, family):
    kind, level, cin, cout = shape
    return f"{kind}_L{level}_{cin}_{cout}_{family}"


def inputs(family, n, cin, cout, K, seed):
    """(X [n, cin], W [K, cin, cout]) float32, deterministic."""
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, cin, generator=g)
    W = torch.randn(K, cin, cout, generator=g) / (cin * K / 2) ** 0.5
    if family == "wide":
        X = X * torch.exp2(torch.randint(-24, 25, X.shape, generator=g).float())
        W = W * torch.exp2(torch.randint(-24, 25, W.shape, generator=g).float())
    elif family == "cancel":
        X[:, 1::2] = -X[:, 0::2]
        u = torch.rand(K, cin // 2, cout, generator=g) * 2 - 1
        W[:, 1::2] = W[:, 0::2] * (1 + u * 2.0 ** -12)
    elif family == "same_sign":
        X, W = X.abs(), W.abs()
    elif family == "tiny":
        X, W = X * 2.0 ** -60, W * 2.0 ** 40
    elif family == "subnormal":
        X, W = X * 2.0 ** -118, W * 2.0 ** 100
    else:
        raise ValueError(family)
    return X.contiguous(), W.contiguous()


def world():
    from agile3d_amd.engine import Scene
    from agile3d_amd.synthetic import make_scene
    from gpu_util import internal_to_oracle_rows
    from oracle import backbone as ob
    coords = make_scene(6000, seed=5)["coords"]
    sc = Scene(torch.from_numpy(coords).cuda())
    lv = ob.SparseLevels(coords)
    maps = [torch.from_numpy(internal_to_oracle_rows(sc, lv, i)) for i in range(5)]
    return sc, lv, maps


def geometry(shape, sc, lv):
    """(rows in, rows out, level out, kernel volume, oracle kernel map)"""
    kind, level, cin, cout = shape
    if kind == "conv3":
        return sc.n[level], sc.n[level], level, 27, lv.kernel_map(level, 3)
    if kind == "down":
        return sc.n[level], sc.n[level + 1], level + 1, 8, lv.stride_map(level)
    kmap = [(rc, rf) for (rf, rc) in lv.stride_map(level - 1)]
    return sc.n[level], sc.n[level - 1], level - 1, 8, kmap


def main(out_path):
    from agile3d_amd import lib as L
    from gpu_util import OneOp, pack_weight
    sc, lv, maps = world()
    kinds = {"conv3": L.OP_CONV3, "down": L.OP_DOWN, "up": L.OP_UP}
    res = {}
    for si, shape in enumerate(SHAPES):
        kind, level, cin, cout = shape
        n_in, n_out, level_out, K, _ = geometry(shape, sc, lv)
        for fi, family in enumerate(FAMILIES):
            X, W = inputs(family, n_in, cin, cout, K, 1000 * si + fi)
            op = OneOp(sc, kinds[kind], level, cin, cout, K, pack_weight(W.cuda()))
            op.buffer(0)[:n_in] = X[maps[level]].cuda()
            op.buffer(1).fill_(float("nan"))
            op.run()
            out = op.buffer(1).cpu()[:n_out]
            back = torch.empty_like(out)
            back[maps[level_out]] = out                   # oracle row order
            res[name_of(shape, family)] = back.numpy()
    np.savez(out_path, **res)
    print("emu_cases: %d outputs, A3D_CONV_EMU=%s" % (len(res), os.environ.get("A3D_CONV_EMU", "0")))


if __name__ == "__main__":
    main(sys.argv[1])
