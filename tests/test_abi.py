"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports every symbol
that include/agile3d_hip.h declares; host-side helpers behave like the ME utilities."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from agile3d_amd import lib
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "agile3d_hip.h")).read()
    declared = set(re.findall(r"\b(a3d_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name)
    assert L.a3d_version() == lib.ABI_VERSION == 3
    # pure host-side queries work without a GPU
    assert L.a3d_scene_workspace_bytes(80000) > 80000 * 27 * 4
    assert L.a3d_decoder_workspace_bytes(80000, 20) > 4 * 80000 * 128 * 4
    assert L.a3d_decoder_workspace_bytes(80000, 257) == 0


def test_struct_layouts_match_header():
    import ctypes as C
    from agile3d_amd import lib
    assert C.sizeof(lib.Op) == 12 * 4 + 3 * 8 + 4 * 4 + 2 * 8 + 2 * 4   # + the fused projection's and the fused head's fields
    assert C.sizeof(lib.BufDesc) == 8
    assert C.sizeof(lib.DecoderLayer) == 29 * 8
    assert C.sizeof(lib.DecoderWeights) == 16 + 8 * 29 * 8 + 11 * 8
    assert C.sizeof(lib.DecoderSample) == 96    # kv0_blocks sits where the struct's tail padding was


def test_sparse_quantize_first_occurrence():
    from agile3d_amd import sparse_quantize, batched_coordinates
    pts = np.array([[0.01, 0.01, 0.01], [0.26, 0.0, 0.0], [0.02, 0.03, 0.04], [0.051, 0.0, 0.0], [0.27, 0.01, 0.0]],
                   np.float32)
    c, idx, inv = sparse_quantize(pts, quantization_size=0.05, return_index=True, return_inverse=True)
    assert c.tolist() == [[0, 0, 0], [5, 0, 0], [1, 0, 0]]
    assert idx.tolist() == [0, 1, 3] and inv.tolist() == [0, 1, 0, 2, 1]
    bc = batched_coordinates([c, c[:2]])
    assert bc.shape == (5, 4) and bc[:, 0].tolist() == [0, 0, 0, 1, 1] and bc.dtype == torch.int32


def test_model_refuses_cpu_forward(full_model_cpu):
    from agile3d_amd import SparseTensor
    x = SparseTensor(features=torch.rand(4, 3), coordinates=torch.tensor([[0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0],
                                                                            [0, 5, 5, 5]], dtype=torch.int32))
    with pytest.raises(RuntimeError):
        full_model_cpu.forward_backbone(x, raw_coordinates=torch.rand(4, 3))


def test_synthetic_scene_is_deterministic_and_unique():
    from agile3d_amd.synthetic import make_clicks, make_scene
    a = make_scene(2000, seed=3)
    b = make_scene(2000, seed=3)
    assert np.array_equal(a["coords"], b["coords"]) and np.array_equal(a["feats"], b["feats"])
    assert len(np.unique(a["coords"], axis=0)) == len(a["coords"])
    assert abs(len(a["coords"]) - 2000) <= 0.05 * 2000
    ci, ct = make_clicks(a["labels"], 3, 2, 1, seed=0)
    assert sorted(ci) == ["0", "1", "2", "3"] and sum(len(v) for v in ci.values()) == 7
    assert sorted(t for v in ct.values() for t in v) == list(range(7))
