"""Helpers for the -m gpu parity tests: run single ops of the backbone program through the C ABI
with explicit inputs, and map between the library's internal row order and the oracle's."""
import ctypes as C

import numpy as np
import torch

from agile3d_amd import lib as L
from agile3d_amd.engine import Scene, _ptr, _stream


def key4(c):
    c = np.asarray(c).astype(np.int64)
    off = 1 << 18   # |xyz| < 2^17 -> every field fits its 20 bits
    return (c[:, 0] << 60) + ((c[:, 1] + off) << 40) + ((c[:, 2] + off) << 20) + (c[:, 3] + off)


def internal_to_oracle_rows(scene: Scene, lv, level):
    """rows[f] = oracle row of internal row f at `level` (oracle = oracle.backbone.SparseLevels)."""
    t = scene.table(level, L.TAB_XYZB).reshape(-1, 4)
    bxyz = np.stack([t[:, 3], t[:, 0], t[:, 1], t[:, 2]], 1)
    oc = lv.levels[level]
    assert len(oc) == len(bxyz), (level, len(oc), len(bxyz))
    ko = key4(oc)
    order = np.argsort(ko)
    pos = np.searchsorted(ko[order], key4(bxyz))
    assert np.array_equal(ko[order][pos], key4(bxyz)), f"level {level}: coordinate sets differ"
    return order[pos]


def pack_weight(w):
    lib = L.load()
    w = w.contiguous()
    K, cin, cout = w.shape
    out = torch.empty(lib.a3d_conv_weight_packed_floats(K, cin, cout), dtype=torch.float32, device=w.device)
    L.check(lib.a3d_pack_conv_weight(_ptr(w), K, cin, cout, _ptr(out), _stream()), "pack")
    return out


class OneOp:
    """A one-op program with caller-filled input / residual buffers."""

    def __init__(self, scene, kind, level_in, cin, cout, kvol, w_packed, scale=None, shift=None, relu=False,
                 use_res=False, in_pad=0, out_pad=0):
        self.scene = scene
        lvl_out = level_in + (1 if kind == L.OP_DOWN else -1 if kind == L.OP_UP else 0)
        self.lvl_in, self.lvl_out = level_in, lvl_out
        self.cin, self.cout, self.in_pad, self.out_pad = cin, cout, in_pad, out_pad
        descs = [(level_in, 4 if kind == L.OP_STEM else cin + in_pad), (lvl_out, cout + out_pad)]
        if use_res:
            descs.append((lvl_out, cout))
        self.descs = descs
        self.bufs = (L.BufDesc * len(descs))(*[L.BufDesc(a, b) for a, b in descs])
        o = L.Op()
        o.kind, o.level_in, o.cin, o.cout = kind, level_in, cin, cout
        o.in_buf, o.in_coff = (L.BUF_NONE, 0) if kind == L.OP_STEM else (0, in_pad)
        o.out_buf, o.out_coff = 1, out_pad
        o.res_buf, o.res_coff = (2, 0) if use_res else (L.BUF_NONE, 0)
        o.relu, o.kernel_volume = int(relu), kvol
        o.w_dev = w_packed.data_ptr()
        o.scale_dev = scale.data_ptr() if scale is not None else None
        o.shift_dev = shift.data_ptr() if shift is not None else None
        self.keep = [w_packed, scale, shift]
        self.ops = (L.Op * 1)(o)
        lib = L.load()
        self.nbytes = lib.a3d_program_workspace_bytes(scene.handle, self.bufs, len(descs), self.ops, 1)
        assert self.nbytes > 0, lib.a3d_last_error()
        self.ws = torch.zeros(self.nbytes, dtype=torch.uint8, device="cuda")

    def buffer(self, i):
        lib = L.load()
        off = lib.a3d_program_buffer_offset(self.scene.handle, self.bufs, len(self.descs), i)
        level, ch = self.descs[i]
        rows = self.scene.n[level] + 1
        return self.ws[off:off + rows * ch * 4].view(torch.float32).view(rows, ch)

    def run(self, feats3=None):
        lib = L.load()
        L.check(lib.a3d_program_run(self.scene.handle, self.bufs, len(self.descs), self.ops, 1,
                                    _ptr(feats3) if feats3 is not None else None, None, 0,
                                    _ptr(self.ws), self.nbytes, _stream()), "a3d_program_run")
        torch.cuda.synchronize()
