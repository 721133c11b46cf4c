"""CPU restatement of the reference's sparse-voxel backbone path (TEST INFRASTRUCTURE).

Follows, line by line:
  * topology           reference ``models/res16unet.py:26-295`` (Res16UNet34C:
                       PLANES ``:371-372``, LAYERS ``:308-310``), ``models/backbone.py:5-7``
  * residual blocks    ``models/modules/resnet_block.py:48-64`` (BasicBlock.forward)
  * block construction ``models/resnet.py:96-149`` (_make_layer: 1x1 projection iff Cin != Cout)
  * layer factories    ``models/modules/common.py:20-31,125-188`` (HYPER_CUBE kernels, no bias)
  * head               ``models/agile3d.py:43-45,179`` (lin_squeeze_head, 1x1 conv + bias)
and MinkowskiEngine's generalized sparse convolution the way its CPU backend
executes it: per kernel offset, gather rows -> dense GEMM -> scatter-add; BatchNorm
and ReLU as separate passes (ME itself is absent -> "parity
unpinned" against ME, pinned against dense conv3d in tests/test_oracle_backbone.py).

Conventions:
  * coordinates int32 [N,4] = (batch, x, y, z); level L holds ``floor(xyz / 2**L)``.
  * odd kernel K, stride 1: offset index k <-> (dx,dy,dz) with x fastest,
    d = (k % K - K//2, (k // K) % K - K//2, k // K**2 - K//2);
    out[u] = sum_k in[u + d_k] @ W[k]           (cross-correlation, W: [K^3, Cin, Cout]).
  * kernel 2, stride 2: out[c] = sum_k in[2c + bits(k)] @ W[k], bits(k) = (k&1, k>>1&1, k>>2&1).
  * transposed kernel 2, stride 2: out_fine[2c + bits(k)] = in[c] @ W[k] on the encoder's
    cached fine coordinate set.
"""
from __future__ import annotations

import numpy as np
import torch

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, reached through ME.MinkowskiBatchNorm (common.py:22)

PLANES = (32, 64, 128, 256, 256, 128, 96, 96)   # res16unet.py:371-372
LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)               # res16unet.py:308-310
INIT_DIM = 32                                   # res16unet.py:14


# ----------------------------------------------------------------------------- coordinates
def _pack(c: np.ndarray) -> np.ndarray:
    """(b,x,y,z) int -> one int64 key (18 signed bits per axis, 10 bits of batch)."""
    c = c.astype(np.int64)
    off = 1 << 17
    return (c[:, 0] << 54) | ((c[:, 1] + off) << 36) | ((c[:, 2] + off) << 18) | (c[:, 3] + off)


class CoordMap:
    """Sorted-key index of one level's coordinates: lookup(coords) -> row or -1."""

    def __init__(self, coords: np.ndarray):
        self.coords = coords
        key = _pack(coords)
        self.order = np.argsort(key, kind="stable")
        self.keys = key[self.order]
        if len(self.keys) > 1 and np.any(self.keys[1:] == self.keys[:-1]):
            raise ValueError("duplicate voxel coordinates")

    def lookup(self, q: np.ndarray) -> np.ndarray:
        qk = _pack(q)
        p = np.searchsorted(self.keys, qk)
        p[p >= len(self.keys)] = len(self.keys) - 1
        hit = self.keys[p] == qk
        rows = np.where(hit, self.order[p], -1)
        return rows


class SparseLevels:
    """The coordinate manager: 5 levels of coordinates + lazily built kernel maps."""

    def __init__(self, coords: np.ndarray, n_levels: int = 5):
        coords = np.ascontiguousarray(coords, dtype=np.int32)
        assert coords.ndim == 2 and coords.shape[1] == 4
        self.levels = [coords]
        self.parent = []  # parent[L][i] = row at level L+1 of fine row i at level L
        for _ in range(1, n_levels):
            fine = self.levels[-1]
            down = fine.copy()
            down[:, 1:] = fine[:, 1:] >> 1  # floor division (arithmetic shift)
            uniq, inv = np.unique(down, axis=0, return_inverse=True)
            self.levels.append(uniq.astype(np.int32))
            self.parent.append(inv.reshape(-1).astype(np.int64))
        self.maps = [CoordMap(c) for c in self.levels]
        self._kmaps = {}

    def n(self, level: int) -> int:
        return len(self.levels[level])

    def kernel_map(self, level: int, ksize: int):
        """list over k of (in_rows, out_rows) for an odd, stride-1 HYPER_CUBE kernel."""
        key = (level, ksize)
        if key not in self._kmaps:
            c = self.levels[level]
            h = ksize // 2
            res = []
            for k in range(ksize ** 3):
                d = np.array([0, k % ksize - h, (k // ksize) % ksize - h, k // (ksize * ksize) - h],
                             dtype=np.int32)
                rows = self.maps[level].lookup(c + d)
                out_rows = np.flatnonzero(rows >= 0)
                res.append((rows[out_rows].astype(np.int64), out_rows.astype(np.int64)))
            self._kmaps[key] = res
        return self._kmaps[key]

    def stride_map(self, level: int):
        """kernel-2 stride-2 map from level -> level+1: list over k of (fine_rows, coarse_rows)."""
        key = (level, "s2")
        if key not in self._kmaps:
            fine = self.levels[level]
            slot = (fine[:, 1] & 1) + 2 * (fine[:, 2] & 1) + 4 * (fine[:, 3] & 1)
            par = self.parent[level]
            res = []
            for k in range(8):
                rows = np.flatnonzero(slot == k).astype(np.int64)
                res.append((rows, par[rows]))
            self._kmaps[key] = res
        return self._kmaps[key]


# ----------------------------------------------------------------------------- layers
def sparse_conv(x: torch.Tensor, W: torch.Tensor, kmap, n_out: int) -> torch.Tensor:
    """ME CPU algorithm: for every kernel offset gather -> GEMM -> scatter-add."""
    out = torch.zeros((n_out, W.shape[-1]), dtype=x.dtype)
    for k, (rin, rout) in enumerate(kmap):
        if len(rin) == 0:
            continue
        rin_t = torch.from_numpy(rin)
        rout_t = torch.from_numpy(rout)
        out.index_add_(0, rout_t, x.index_select(0, rin_t) @ W[k])
    return out


# the activation of the backbone: a module attribute so that a test can substitute "multiply by a given 0/1 mask" for
# the gradient comparison of the training path (ReLU's derivative jumps at 0: with the masks of the implementation under
# test, two forward passes that differ in the last bits still define the SAME piecewise-linear function)
RELU = torch.relu


def batch_norm_eval(x, sd, prefix):
    """nn.BatchNorm1d in eval mode over the [N,C] rows (ME.MinkowskiBatchNorm, common.py:22)."""
    w, b = sd[prefix + "bn.weight"], sd[prefix + "bn.bias"]
    m, v = sd[prefix + "bn.running_mean"], sd[prefix + "bn.running_var"]
    return torch.nn.functional.batch_norm(x, m, v, w, b, training=False, eps=BN_EPS)


def _k3(W):
    return W if W.dim() == 3 else W.unsqueeze(0)


def batch_norm_train(x, sd, prefix, momentum=None):
    """nn.BatchNorm1d in TRAINING mode (batch statistics over all rows of the batch; the running statistics in ``sd``
    are updated in place like the module's buffers) -- the training path, engine.py:26-179.  Momentum as the reference
    builds its layers: the two norms inside a BasicBlock keep the class default 0.1 (``_make_layer`` does not pass
    ``bn_momentum`` to the blocks, resnet.py:123-147, resnet_block.py:19), every other norm gets config.bn_momentum =
    0.02 (res16unet.py:29,49; the projection's norm: resnet.py:116-121)."""
    if momentum is None:
        momentum = 0.1 if prefix.endswith(("norm1.", "norm2.")) else 0.02
    return torch.nn.functional.batch_norm(x, sd[prefix + "bn.running_mean"], sd[prefix + "bn.running_var"],
                                          sd[prefix + "bn.weight"], sd[prefix + "bn.bias"], training=True,
                                          momentum=momentum, eps=BN_EPS)


def basic_block(x, sd, prefix, lv: SparseLevels, level: int, bn=batch_norm_eval):
    """BasicBlock.forward, resnet_block.py:48-64."""
    km = lv.kernel_map(level, 3)
    n = lv.n(level)
    out = sparse_conv(x, sd[prefix + "conv1.kernel"], km, n)
    out = RELU(bn(out, sd, prefix + "norm1."))
    out = sparse_conv(out, sd[prefix + "conv2.kernel"], km, n)
    out = bn(out, sd, prefix + "norm2.")
    if (prefix + "downsample.0.kernel") in sd:
        Wp = sd[prefix + "downsample.0.kernel"]
        Wp = Wp if Wp.dim() == 2 else Wp[0]
        residual = bn(x @ Wp, sd, prefix + "downsample.1.")
    else:
        residual = x
    return RELU(out + residual)


def _layer(x, sd, prefix, n_blocks, lv, level, bn=batch_norm_eval):
    for i in range(n_blocks):
        x = basic_block(x, sd, f"{prefix}{i}.", lv, level, bn)
    return x


def res16unet34c_forward(sd, lv: SparseLevels, feats: torch.Tensor, prefix: str = "backbone.", bn=batch_norm_eval):
    """Res16UNetBase.forward, res16unet.py:222-295.  Returns (out [N0,96], feature_maps[5]).  ``bn`` selects eval-mode
    (inference path) or ``batch_norm_train`` (training path; differentiable through torch autograd)."""
    norm = bn              # (down / up below take the NAME of their norm layer in a parameter called bn)
    p = prefix
    ksz = round(sd[p + "conv0p1s1.kernel"].shape[0] ** (1 / 3))
    fm = []
    out = sparse_conv(feats, sd[p + "conv0p1s1.kernel"], lv.kernel_map(0, ksz), lv.n(0))
    out_p1 = RELU(norm(out, sd, p + "bn0."))

    def down(x, conv, bn, level):
        y = sparse_conv(x, sd[p + conv + ".kernel"], lv.stride_map(level), lv.n(level + 1))
        return RELU(norm(y, sd, p + bn + "."))

    def up(x, conv, bn, level_out):
        # transposed conv = the stride map of level_out with in/out swapped (App. B.5)
        kmap = [(rc, rf) for (rf, rc) in lv.stride_map(level_out)]
        y = sparse_conv(x, sd[p + conv + ".kernel"], kmap, lv.n(level_out))
        return RELU(norm(y, sd, p + bn + "."))

    out = down(out_p1, "conv1p1s2", "bn1", 0)
    out_b1p2 = _layer(out, sd, p + "block1.", LAYERS[0], lv, 1, bn)
    out = down(out_b1p2, "conv2p2s2", "bn2", 1)
    out_b2p4 = _layer(out, sd, p + "block2.", LAYERS[1], lv, 2, bn)
    out = down(out_b2p4, "conv3p4s2", "bn3", 2)
    out_b3p8 = _layer(out, sd, p + "block3.", LAYERS[2], lv, 3, bn)
    out = down(out_b3p8, "conv4p8s2", "bn4", 3)
    out = _layer(out, sd, p + "block4.", LAYERS[3], lv, 4, bn)
    fm.append(out)

    out = up(out, "convtr4p16s2", "bntr4", 3)
    out = torch.cat([out, out_b3p8], 1)            # me.cat(out, skip): skip = LAST columns
    out = _layer(out, sd, p + "block5.", LAYERS[4], lv, 3, bn)
    fm.append(out)
    out = up(out, "convtr5p8s2", "bntr5", 2)
    out = torch.cat([out, out_b2p4], 1)
    out = _layer(out, sd, p + "block6.", LAYERS[5], lv, 2, bn)
    fm.append(out)
    out = up(out, "convtr6p4s2", "bntr6", 1)
    out = torch.cat([out, out_b1p2], 1)
    out = _layer(out, sd, p + "block7.", LAYERS[6], lv, 1, bn)
    fm.append(out)
    out = up(out, "convtr7p2s2", "bntr7", 0)
    out = torch.cat([out, out_p1], 1)
    out = _layer(out, sd, p + "block8.", LAYERS[7], lv, 0, bn)
    fm.append(out)
    return out, fm


def forward_backbone(sd, coords: np.ndarray, feats: torch.Tensor, raw_xyz: torch.Tensor,
                     levels: SparseLevels | None = None):
    """Agile3d.forward_backbone, agile3d.py:163-181 (batch size 1 per call, as the reference's
    CPU path requires).  Returns dict with pcd_features [N,128], pos_enc [N,128] (the only
    level the decoder reads: hlevels=[4], agile3d.py:278), out96, feature_maps, levels."""
    from .decoder import fourier_pos_enc
    with torch.no_grad():
        lv = levels if levels is not None else SparseLevels(coords)
        out, fm = res16unet34c_forward(sd, lv, feats)
        Wh = sd["lin_squeeze_head.kernel"]
        Wh = Wh if Wh.dim() == 2 else Wh[0]
        pcd = out @ Wh + sd["lin_squeeze_head.bias"].reshape(1, -1)
        pos = fourier_pos_enc(raw_xyz, sd["pos_enc.gauss_B"], raw_xyz.min(0)[0], raw_xyz.max(0)[0])
    return {"pcd_features": pcd, "pos_enc": pos, "out96": out, "feature_maps": fm, "levels": lv}
