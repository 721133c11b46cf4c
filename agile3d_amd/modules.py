"""Parameter containers that reproduce the reference's ``state_dict`` key layout.

The reference's model is an ``nn.Module`` tree whose leaves are MinkowskiEngine layers
(``.kernel`` / ``.bias`` parameters, ``MinkowskiBatchNorm.bn``) and torch layers
(``nn.MultiheadAttention`` ...).  Checkpoints (``main.py:190-202``) are plain
``state_dict``s of that tree, so a drop-in replacement has to own identically named
parameters ("checkpoint / resume"; 455 entries, 39 289 760 params).

None of these containers computes anything: the arithmetic runs in the HIP library
(``agile3d_amd/csrc``) through ``agile3d_amd.engine``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class SparseConvParams(nn.Module):
    """Holds what ``ME.MinkowskiConvolution(Transpose)`` holds: ``kernel`` [K^3, Cin, Cout]
    ([Cin, Cout] when the kernel volume is 1) and optionally ``bias`` [1, Cout].
    Reference factory: ``models/modules/common.py:125-188``."""

    def __init__(self, cin, cout, kernel_volume, bias=False, transposed=False, stride=1):
        super().__init__()
        self.cin, self.cout, self.kernel_volume = cin, cout, kernel_volume
        self.transposed, self.stride = transposed, stride
        shape = (cin, cout) if kernel_volume == 1 else (kernel_volume, cin, cout)
        self.kernel = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(1, cout)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # ME's reset_parameters: uniform(-1/sqrt(n), 1/sqrt(n)) with n = in_channels * kernel_volume for a convolution
        # and out_channels * kernel_volume for a transposed one (is_transpose=True); it matters when training from
        # scratch (inference weights come from the checkpoint)
        n = (self.cout if self.transposed else self.cin) * self.kernel_volume
        std = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-std, std)
            if self.bias is not None:
                self.bias.uniform_(-std, std)

    def kernel3(self):
        return self.kernel if self.kernel.dim() == 3 else self.kernel.unsqueeze(0)


class SparseBatchNormParams(nn.Module):
    """``ME.MinkowskiBatchNorm`` wraps ``nn.BatchNorm1d`` under the attribute ``bn``
    (state-dict prefix ``.bn.``; reference ``common.py:20-22``, ``resnet.py:90-94``)."""

    def __init__(self, c, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(c, momentum=momentum)


class BasicBlockParams(nn.Module):
    """Parameters of ``BasicBlock`` (``resnet_block.py:7-64``): conv1,norm1,conv2,norm2 and the
    optional ``downsample`` = Sequential(1x1 conv, BN) (``resnet.py:107-123``)."""

    def __init__(self, cin, cout, bn_momentum_proj=0.1):
        super().__init__()
        self.conv1 = SparseConvParams(cin, cout, 27)
        self.norm1 = SparseBatchNormParams(cout)            # BasicBlock BNs keep momentum 0.1
        self.conv2 = SparseConvParams(cout, cout, 27)
        self.norm2 = SparseBatchNormParams(cout)
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(SparseConvParams(cin, cout, 1),
                                            SparseBatchNormParams(cout, bn_momentum_proj))


def _make_layer(cin, cout, n_blocks, bn_momentum):
    blocks = [BasicBlockParams(cin, cout, bn_momentum)]
    blocks += [BasicBlockParams(cout, cout, bn_momentum) for _ in range(1, n_blocks)]
    return nn.Sequential(*blocks)


class Res16UNet34CParams(nn.Module):
    """Parameter tree of ``Res16UNet34C(3, 20, config, out_fpn=True)``
    (``models/res16unet.py:26-220,371-372``; ``models/backbone.py:5-7``)."""

    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)
    INIT_DIM = 32

    def __init__(self, in_channels=3, conv1_kernel_size=5, bn_momentum=0.02):
        super().__init__()
        P, L, m = self.PLANES, self.LAYERS, bn_momentum
        self.conv1_kernel_size = conv1_kernel_size
        self.conv0p1s1 = SparseConvParams(in_channels, 32, conv1_kernel_size ** 3)
        self.bn0 = SparseBatchNormParams(32, m)
        self.conv1p1s2 = SparseConvParams(32, 32, 8, stride=2)
        self.bn1 = SparseBatchNormParams(32, m)
        self.block1 = _make_layer(32, P[0], L[0], m)
        self.conv2p2s2 = SparseConvParams(P[0], P[0], 8, stride=2)
        self.bn2 = SparseBatchNormParams(P[0], m)
        self.block2 = _make_layer(P[0], P[1], L[1], m)
        self.conv3p4s2 = SparseConvParams(P[1], P[1], 8, stride=2)
        self.bn3 = SparseBatchNormParams(P[1], m)
        self.block3 = _make_layer(P[1], P[2], L[2], m)
        self.conv4p8s2 = SparseConvParams(P[2], P[2], 8, stride=2)
        self.bn4 = SparseBatchNormParams(P[2], m)
        self.block4 = _make_layer(P[2], P[3], L[3], m)
        self.convtr4p16s2 = SparseConvParams(P[3], P[4], 8, transposed=True, stride=2)
        self.bntr4 = SparseBatchNormParams(P[4], m)
        self.block5 = _make_layer(P[4] + P[2], P[4], L[4], m)
        self.convtr5p8s2 = SparseConvParams(P[4], P[5], 8, transposed=True, stride=2)
        self.bntr5 = SparseBatchNormParams(P[5], m)
        self.block6 = _make_layer(P[5] + P[1], P[5], L[5], m)
        self.convtr6p4s2 = SparseConvParams(P[5], P[6], 8, transposed=True, stride=2)
        self.bntr6 = SparseBatchNormParams(P[6], m)
        self.block7 = _make_layer(P[6] + P[0], P[6], L[6], m)
        self.convtr7p2s2 = SparseConvParams(P[6], P[7], 8, transposed=True, stride=2)
        self.bntr7 = SparseBatchNormParams(P[7], m)
        self.block8 = _make_layer(P[7] + 32, P[7], L[7], m)


class _AttnLayerParams(nn.Module):
    """CrossAttentionLayer / SelfAttentionLayer parameters (``attention_block.py:5-24,63-82``)."""

    def __init__(self, d_model, nhead, attn_name):
        super().__init__()
        setattr(self, attn_name, nn.MultiheadAttention(d_model, nhead, dropout=0.0))
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class FFNLayerParams(nn.Module):
    """FFNLayer parameters (``attention_block.py:126-147``)."""

    def __init__(self, d_model, dim_feedforward):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class FourierPosEncParams(nn.Module):
    """``PositionEmbeddingCoordsSine(pos_type='fourier')``: owns the random ``gauss_B`` buffer
    (``position_embedding.py:67-73``) -- it must come from the checkpoint."""

    def __init__(self, d_pos=128, d_in=3, gauss_scale=1.0):
        super().__init__()
        B = torch.empty((d_in, d_pos // 2)).normal_() * gauss_scale
        self.register_buffer("gauss_B", B)


def cross_attention_params(d_model, nhead):
    return _AttnLayerParams(d_model, nhead, "multihead_attn")


def self_attention_params(d_model, nhead):
    return _AttnLayerParams(d_model, nhead, "self_attn")
