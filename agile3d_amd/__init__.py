"""agile3d_amd -- MI355X-native implementation of the AGILE3D model hot path
(forward_backbone + forward_mask) behind the reference's own model API.

Public surface (mirrors the reference):
    build_model(args)            models/__init__.py:5-6
    Agile3d.forward_backbone     models/agile3d.py:163-181
    Agile3d.forward_mask         models/agile3d.py:183-339
    SparseTensor, utils          the subset of MinkowskiEngine the callers touch
Around the path (same names as the reference): ``datasets`` (scan datasets + collate), ``ply`` (binary PLY),
``evaluate`` (Evaluate loops, NoC / IoU@k tables), ``clicks`` (click simulator, IoU), ``criterion`` (mask losses).
"""
from .hostcpu import cap_host_threads
cap_host_threads()       # torch's CPU pool sized to the container's CPU quota (hostcpu.py: an oversized pool freezes the launch thread)
from .model import build_model, build_agile3d, Agile3d, default_args, randomize_bn_stats  # noqa: F401
from .sparse import SparseTensor, sparse_quantize, batched_coordinates  # noqa: F401
from . import utils  # noqa: F401  (ME.utils.sparse_quantize / ME.utils.batched_coordinates)


def build_criterion(args):
    """models/__init__.py:10-11."""
    from .criterion import build_mask_criterion
    return build_mask_criterion(args)

__all__ = ["build_model", "build_agile3d", "Agile3d", "default_args", "randomize_bn_stats",
           "SparseTensor", "sparse_quantize", "batched_coordinates", "utils", "build_criterion"]
