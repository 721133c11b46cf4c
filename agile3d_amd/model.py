"""``build_model(args)`` -> ``Agile3d`` with ``forward_backbone`` / ``forward_mask``.

Drop-in for the reference's model boundary:
  * ``models/__init__.py:5-6``      build_model(args)
  * ``models/agile3d.py:163-181``   forward_backbone(x, raw_coordinates)
  * ``models/agile3d.py:183-339``   forward_mask(pcd_features, aux, coordinates,
                                    pos_encodings_pcd, click_idx, click_time_idx)
Same argument meaning, same return structures, same ``state_dict`` keys.  All arithmetic
runs in ``libagile3d_hip.so`` (hand-written gfx950 kernels); there is NO CPU fallback --
calling a forward without the HIP library or on a non-GPU tensor raises.  Both modes of the ``nn.Module`` are
served: ``eval()`` by the fused inference kernels, ``train()`` by the training path tied into torch.autograd.
"""
from __future__ import annotations

import argparse

import torch
import torch.nn as nn

from . import modules as M


def default_args(**overrides):
    """The model-relevant flag defaults of the reference's entry points
    (``main.py:24-84``, ``eval_multi_obj.py:28-72``)."""
    a = argparse.Namespace(
        conv1_kernel_size=5, bn_momentum=0.02, voxel_size=0.05, hidden_dim=128,
        dim_feedforward=1024, num_heads=8, num_decoders=3, num_bg_queries=10, dropout=0.0,
        pre_norm=False, normalize_pos_enc=True, positional_encoding_type="fourier",
        gauss_scale=1.0, hlevels=[4], shared_decoder=False, aux=True)
    for k, v in overrides.items():
        setattr(a, k, v)
    return a


def kernel_order_permutation(kernel_volume: int):
    """perm[k_x_fastest] = k_z_fastest for a cubic kernel of ``kernel_volume`` = s^3 offsets: the library (and
    oracle/backbone.py) enumerate kernel offsets with x fastest, k = ix + s iy + s^2 iz (from
    memory of MinkowskiEngine v0.5 -- its source is not available offline); a checkpoint written by a build that
    enumerates z fastest holds the weight of offset (ix, iy, iz) at k' = iz + s iy + s^2 ix."""
    s = round(kernel_volume ** (1.0 / 3.0))
    if s * s * s != kernel_volume:
        return None
    perm = []
    for k in range(kernel_volume):
        ix, iy, iz = k % s, (k // s) % s, k // (s * s)
        perm.append(iz + s * iy + s * s * ix)
    return perm


def convert_kernel_order(state_dict, kernel_order: str):
    """A copy of ``state_dict`` whose sparse-conv kernels ([K, Cin, Cout], K = 8 / 27 / 125) are re-indexed from
    ``kernel_order`` ("x_fastest" = the library's own order: no change; "z_fastest") to the library's order.  The load-time
    switch SURVEY.md section 7 ("Hard parts") asks for: which enumeration the authors' checkpoint uses can only be told
    by IoU@5 on ScanNet once weights and data are supplied; random-weight parity does not depend on it."""
    if kernel_order in (None, "", "x_fastest"):
        return dict(state_dict)
    if kernel_order != "z_fastest":
        raise ValueError(f"kernel_order must be 'x_fastest' or 'z_fastest', not {kernel_order!r}")
    out = {}
    for k, v in state_dict.items():
        if k.endswith(".kernel") and torch.is_tensor(v) and v.dim() == 3 and v.shape[0] > 1:
            perm = kernel_order_permutation(v.shape[0])
            if perm is not None:
                v = v[torch.tensor(perm, dtype=torch.long, device=v.device)]
        out[k] = v
    return out


class Agile3d(nn.Module):
    """Parameter layout of ``models/agile3d.py:19-138``; compute in HIP."""

    def __init__(self, args):
        super().__init__()
        if args.pre_norm:
            raise NotImplementedError("pre_norm=True is outside the hot path (reference default False)")
        if args.positional_encoding_type != "fourier":
            raise NotImplementedError("only the default 'fourier' position encoding is on the hot path")
        if list(args.hlevels) != [4]:
            raise NotImplementedError("only hlevels=[4] (reference default) is on the hot path")
        if args.dropout != 0.0:
            raise NotImplementedError("dropout != 0 is outside the hot path (reference default 0.0)")
        if args.hidden_dim != 128 or args.num_heads != 8:
            raise NotImplementedError("kernels are specialised for hidden_dim=128, num_heads=8")
        self.args = args
        d, h, ff = args.hidden_dim, args.num_heads, args.dim_feedforward
        self.mask_dim = d
        self.num_heads = h
        self.num_decoders = args.num_decoders
        self.num_bg_queries = args.num_bg_queries
        self.shared_decoder = args.shared_decoder
        self.hlevels = list(args.hlevels)
        self.aux = args.aux
        self.voxel_size = args.voxel_size

        self.backbone = M.Res16UNet34CParams(3, args.conv1_kernel_size, args.bn_momentum)
        self.lin_squeeze_head = M.SparseConvParams(self.backbone.PLANES[7], d, 1, bias=True)
        self.bg_query_feat = nn.Embedding(args.num_bg_queries, d)
        self.bg_query_pos = nn.Embedding(args.num_bg_queries, d)
        self.mask_embed_head = nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Linear(d, d))
        self.pos_enc = M.FourierPosEncParams(d, 3, args.gauss_scale)

        n_shared = 1 if args.shared_decoder else args.num_decoders
        self.c2s_attention = nn.ModuleList()
        self.c2c_attention = nn.ModuleList()
        self.ffn_attention = nn.ModuleList()
        self.s2c_attention = nn.ModuleList()
        for _ in range(n_shared):
            self.c2s_attention.append(nn.ModuleList([M.cross_attention_params(d, h)]))
            self.s2c_attention.append(nn.ModuleList([M.cross_attention_params(d, h)]))
            self.c2c_attention.append(nn.ModuleList([M.self_attention_params(d, h)]))
            self.ffn_attention.append(nn.ModuleList([M.FFNLayerParams(d, ff)]))
        self.decoder_norm = nn.LayerNorm(d)
        self._engine = None          # lazily created agile3d_amd.engine.Engine
        self._packed_version = None

    def train(self, mode: bool = True):
        """``nn.Module.train`` walks the module tree recursively (~900 Python calls, 0.4 ms for this model); the training step
        flips the mode three times per iteration (engine.py:53,84,118 of the reference), twice while the device has nothing
        queued.  The flat list of modules is kept; the tree is fixed after ``__init__`` (``_flat_modules = None`` after
        changing it by hand)."""
        if not isinstance(mode, bool):
            raise ValueError("training mode is expected to be boolean")
        flat = self.__dict__.get("_flat_modules")
        if flat is None:
            flat = list(self.modules())
            self.__dict__["_flat_modules"] = flat
        for m in flat:
            m.training = mode
        return self

    # ------------------------------------------------------------------ engine plumbing
    def _get_engine(self):
        from .engine import Engine
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("agile3d_amd: the model must live on a ROCm GPU (model.to('cuda')); "
                               "there is no CPU path")
        if self._engine is None or self._engine.device != dev:
            self._engine = Engine(self, dev)
        return self._engine

    def load_state_dict(self, state_dict, strict=True, kernel_order=None, **kw):
        """``kernel_order``: enumeration of the kernel offsets in the FILE ("x_fastest" = the library's own, default;
        "z_fastest": see ``convert_kernel_order``).  The permutation is applied only when the caller passes the
        argument: state dicts this library wrote (``save_checkpoint``, ``model_a.state_dict()``) are already in its
        own order, so a sticky default (args / environment) would permute them a second time on every resume.
        Importing a foreign checkpoint is the explicit step ``import_state_dict``."""
        sd = convert_kernel_order(state_dict, kernel_order)
        own = self.state_dict()   # accept ME<0.5 checkpoints holding 1x1 kernels as [1,Cin,Cout]
        for k, v in list(sd.items()):
            if k in own and own[k].dim() == 2 and v.dim() == 3 and v.shape[0] == 1 and k.endswith(".kernel"):
                sd[k] = v[0]
        res = super().load_state_dict(sd, strict=strict, **kw)
        if self._engine is not None:
            self._engine.mark_stale()
        return res

    def import_state_dict(self, state_dict, strict=False, kernel_order=None):
        """Load a checkpoint written by the REFERENCE (``ckpt['model']``, eval_multi_obj.py:60-63): the one place where
        ``args.kernel_order`` / A3D_KERNEL_ORDER are consulted for the enumeration of kernel offsets in the file."""
        import os
        order = kernel_order or getattr(self.args, "kernel_order", None) or os.environ.get("A3D_KERNEL_ORDER")
        return self.load_state_dict(state_dict, strict=strict, kernel_order=order)

    # ------------------------------------------------------------------ the hot path
    def forward_backbone(self, x, raw_coordinates=None):
        """Reference ``agile3d.py:163-181``.  ``x``: SparseTensor of int32 [N,4] coords +
        fp32 [N,3] colours; ``raw_coordinates`` fp32 [N,3].  Returns
        (pcd_features, aux, coordinates, pos_encodings_pcd), opaque to callers.

        ``model.eval()``: the inference kernels (BatchNorm folded from the running statistics).  ``model.train()``
        (engine.py:38,53 of the reference): BatchNorm on the batch statistics, activations kept, the result part of
        torch's autograd graph -- ``losses.backward()`` fills ``.grad`` of every parameter (agile3d_amd/autograd.py)."""
        eng = self._get_engine()
        if self.training:
            return eng.forward_backbone_train(x, raw_coordinates)
        eng.refresh_weights_if_stale()
        return eng.forward_backbone(x, raw_coordinates)

    def forward_mask(self, pcd_features, aux, coordinates, pos_encodings_pcd,
                     click_idx=None, click_time_idx=None):
        """Reference ``agile3d.py:183-339``.  Accepts the results of either mode's ``forward_backbone`` (the training
        loop runs its no-grad click rounds in eval mode on the training-mode backbone's features, engine.py:82-115)."""
        eng = self._get_engine()
        if self.training:
            return eng.forward_mask_train(pcd_features, aux, coordinates, pos_encodings_pcd, click_idx, click_time_idx)
        return eng.forward_mask(pcd_features, aux, coordinates, pos_encodings_pcd, click_idx, click_time_idx)


def build_agile3d(args):
    return Agile3d(args)


def build_model(args):
    """Reference ``models/__init__.py:5-6``."""
    return build_agile3d(args)


def randomize_bn_stats(model: nn.Module, seed: int = 0):
    """Give every BatchNorm non-trivial running statistics / affine so that eval-mode folding is
    exercised (mean N(0,0.1), var U(0.5,1.5))."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    return model
