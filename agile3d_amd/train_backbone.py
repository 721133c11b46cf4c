"""Training-mode forward and backward of the backbone (Res16UNet34C + lin_squeeze_head) on the HIP kernels -- the
backbone half of SURVEY.md section 8 row f-2 (``engine.py:26-179``: ``model.train()``, ``losses.backward()``).

    tape = BackboneTape(model, scene, feats3)        forward: res16unet.py:222-295 with BatchNorm on batch statistics
    pcd = tape.output                                [N, 128] in the caller's row order (agile3d.py:179)
    grads = tape.backward(d_pcd)                     {state-dict key: gradient} for every backbone parameter

Every FLOP runs in libagile3d_hip (conv forward = the inference kernel on the caller's buffers, a3d_conv_apply; conv input
gradient = the same kernel on the transposed maps; k_wgrad; the BatchNorm training kernels; k_stem_wgrad); this module is
the reverse-mode bookkeeping: which activation feeds which layer, the two-way fan-outs of the residual / skip
connections (a tensor add), the channel split of the concatenations.  Activations and gradients carry the zero row the
conv kernels gather for a missing neighbour ([n + 1, C] tensors, written by the producing kernel), both orientations of
every conv weight are packed once per weight version (``PackedWeights``), workspaces come from one arena.  The decoder's
half is ``train_decoder.DecoderTape``, the whole iteration ``train_step.train_one_step``.
"""
from __future__ import annotations

import torch

from . import backward as B
from . import lib as L


class _T:
    """Activation node: value [n + 1, C] (row n = the zero row), level, accumulated gradient (same shape)."""
    __slots__ = ("v", "level", "g")

    def __init__(self, v, level):
        self.v, self.level, self.g = v, level, None

    def add_grad(self, g):
        self.g = g if self.g is None else self.g + g


class _View:
    """What the tests read off ``relu_levels``: the [n, C] rows of an activation."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


class PackedWeights:
    """Both orientations of every sparse-conv kernel in the MFMA fragment order, packed once per weight version: the
    forward weight and the slices of the input-gradient conv's W' (backward.packed_input_grad_weights).  A tape asks
    ``get(conv)``; entries are refreshed when the parameter tensor was written (torch bumps ``_version``), after a step of
    the library's own AdamW (it writes through raw pointers and bumps ``optim.WEIGHT_EPOCH``) or after ``invalidate()``.

    ``refresh_all()`` (called when a tape starts) repacks EVERY known entry that went stale with one launch of
    ``a3d_pack_conv_weights_multi`` into the buffers the entries already own: after an optimiser step that is all of them
    -- one launch instead of ~230 packs plus the transposes / flips / slice copies that fed them."""

    def __init__(self):
        self._c = {}             # id(param) -> [version, packed forward, input-grad parts, conv, kind]
        self.epoch = 0
        self._table = None       # (device job table, n_jobs, n_chunks, signature)

    def invalidate(self):
        self.epoch += 1

    def _version(self, conv):
        from .optim import WEIGHT_EPOCH
        return (int(conv.kernel._version), self.epoch, WEIGHT_EPOCH[0], conv.kernel.data_ptr())

    def get(self, conv, kind):
        key = id(conv.kernel)
        ver = self._version(conv)
        hit = self._c.get(key)
        if hit is None or hit[0] != ver:
            w = conv.kernel3().detach().contiguous()
            hit = self._c[key] = [ver, B.pack_weight(w), B.packed_input_grad_weights(kind, w), conv, kind]
            self._table = None
        return hit[1], hit[2]

    def _build_table(self):
        import numpy as np
        lib = L.load()
        dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("K", "<i4"), ("cin", "<i4"), ("cout", "<i4"), ("src_cin", "<i4"),
                       ("src_cout", "<i4"), ("transposed", "<i4"), ("flip", "<i4"), ("c0", "<i4"), ("chunk0", "<i4"),
                       ("pad", "<i4")])
        assert dt.itemsize == 56
        rows, chunk, sig, dev = [], 0, [], None
        for key, (ver, wf, parts, conv, kind) in self._c.items():
            w = conv.kernel3().detach()
            K, cin, cout = w.shape
            if not w.is_contiguous() or lib.a3d_conv_weight_packed_floats(K, cin, cout) != K * cin * cout:
                return None          # a weight the one-launch pack does not cover (emulated-fp32 build, strided view)
            dev = w.device
            jobs = [(w.data_ptr(), wf.data_ptr(), K, cin, cout, cin, cout, 0, 0, 0)]
            for c0, width, pk in parts:
                if lib.a3d_conv_weight_packed_floats(K, cout, width) != K * cout * width:
                    return None
                jobs.append((w.data_ptr(), pk.data_ptr(), K, cout, width, cin, cout, 1, 1 if kind == L.OP_CONV3 else 0, c0))
            for j in jobs:
                rows.append(j + (chunk, 0))
                chunk += (j[2] * j[3] * j[4] + 4095) // 4096
            sig.append((key, w.data_ptr()))
        if not rows:
            return None
        tab = np.array(rows, dtype=dt)
        return (torch.from_numpy(tab.view(np.uint8)).to(dev), len(rows), chunk, tuple(sig))

    def refresh_all(self):
        stale = [h for h in self._c.values() if h[0] != self._version(h[3])]
        if not stale:
            return
        sig = tuple((k, h[3].kernel3().data_ptr()) for k, h in self._c.items())
        if self._table is None or self._table[3] != sig:
            self._table = self._build_table()
        if self._table is None:
            return                   # get() repacks entry by entry
        tab, n_jobs, n_chunks, _ = self._table
        L.check(L.load().a3d_pack_conv_weights_multi(tab.data_ptr(), n_jobs, n_chunks, B._stream()),
                "a3d_pack_conv_weights_multi")
        for h in self._c.values():
            h[0] = self._version(h[3])


def packed_weights_of(model) -> PackedWeights:
    pw = getattr(model, "_a3d_packed_train", None)
    if pw is None:
        pw = PackedWeights()
        object.__setattr__(model, "_a3d_packed_train", pw)
    return pw


def _run_stem(scene, w, feats3, kvol):
    """conv0p1s1 on its own (OP_STEM reads the caller-ordered features) -> [n0 + 1, 32] internal order, zero row last."""
    lib = L.load()
    n0 = scene.n[0]
    bufs = (L.BufDesc * 1)(L.BufDesc(0, 32))
    o = L.Op()
    o.kind, o.level_in, o.cin, o.cout = L.OP_STEM, 0, 3, 32
    o.in_buf, o.in_coff, o.out_buf, o.out_coff = L.BUF_NONE, 0, 0, 0
    o.res_buf, o.res_coff, o.relu, o.kernel_volume = L.BUF_NONE, 0, 0, kvol
    o.w_dev, o.scale_dev, o.shift_dev = w.data_ptr(), None, None
    ops = (L.Op * 1)(o)
    nbytes = lib.a3d_program_workspace_bytes(scene.handle, bufs, 1, ops, 1)
    ws = B._workspace(nbytes, feats3.device, "stem")
    L.check(lib.a3d_program_run(scene.handle, bufs, 1, ops, 1, B._ptr(feats3), None, 0, B._ptr(ws), nbytes, B._stream()),
            "a3d_program_run")
    off = lib.a3d_program_buffer_offset(scene.handle, bufs, 1, 0)
    return ws[off:off + (n0 + 1) * 32 * 4].view(torch.float32).view(n0 + 1, 32).clone()


def _sync_bn_default():
    """SyncBN is opt-in: ``A3D_SYNC_BN=1`` (or ``BackboneTape(..., sync_bn=True)``) and an initialised process group of
    more than one rank.  Off, every rank normalises over its own scenes (what plain DDP does)."""
    import os

    import torch.distributed as dist
    return os.environ.get("A3D_SYNC_BN", "0") == "1" and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class BackboneTape:
    def __init__(self, model, scene, feats3: torch.Tensor, sync_bn=None):
        if not feats3.is_cuda:
            raise RuntimeError("BackboneTape runs on the GPU only")
        self.model, self.scene = model, scene
        self.sync_bn = _sync_bn_default() if sync_bn is None else bool(sync_bn)
        self.feats3 = feats3.to(torch.float32).contiguous()
        self.steps = []          # backward closures, in forward order
        self.relu_levels = []
        self.grads = {}
        self._names = {id(p): n for n, p in model.named_parameters()}
        self.packed = packed_weights_of(model)
        self.packed.refresh_all()
        self._bns = []           # the BatchNorms this forward pass ran through (running statistics updated in place)
        self._forward()
        # the kernels wrote running_mean / running_var through raw device pointers: torch's version counters did not
        # move, so (1) count the batch like nn.BatchNorm1d does (state-dict parity with torch, one multi-tensor launch;
        # the buffers' version bump is what refresh_weights_if_stale(check_versions=True) sees) and (2) tell the
        # inference engine directly that its folded backbone program is out of date
        if self._bns:
            torch._foreach_add_([b.num_batches_tracked for b in self._bns], 1)
        eng = getattr(model, "_engine", None)
        if eng is not None:
            eng._stale = True

    # ------------------------------------------------------------------ layers
    def _pgrad(self, param, g):
        name = self._names[id(param)]
        g = g.reshape(param.shape)
        self.grads[name] = g if name not in self.grads else self.grads[name] + g

    def _conv(self, kind, x: _T, conv) -> _T:
        K, cin, cout = conv.kernel3().shape
        wf, wb = self.packed.get(conv, kind)
        sc = self.scene
        y = _T(B.conv_apply(sc, kind, x.level, wf, x.v, cin, cout), B.level_out(kind, x.level))
        n_in, n_out = sc.n[x.level], sc.n[y.level]

        def back():
            self._pgrad(conv.kernel, B.conv_weight_grad(sc, kind, x.level, x.v[:n_in], y.g[:n_out]))
            x.add_grad(B.conv_input_grad_apply(sc, kind, x.level, wb, y.g, cin, cout))
        self.steps.append(back)
        return y

    def _bn(self, x: _T, norm, res: _T | None = None, relu=True) -> _T:
        b = norm.bn
        self._bns.append(b)
        n = self.scene.n[x.level]
        xv, rv = x.v[:n], (res.v[:n] if res is not None else None)
        n_glob = None
        if self.sync_bn:
            v, mean, rstd, n_glob = B.bn_sync_forward(xv, b.weight.detach(), b.bias.detach(), b.eps, rv, relu, b.running_mean,
                                                      b.running_var, b.momentum, zero_row=True)
        else:
            v, mean, rstd = B.bn_train_forward(xv, b.weight.detach(), b.bias.detach(), b.eps, rv, relu, b.running_mean,
                                               b.running_var, b.momentum, zero_row=True)
        y = _T(v, x.level)
        if relu:
            self.relu_levels.append((x.level, _View(v[:n])))   # forward order of the ReLUs (tests read the 0/1 masks)

        def back():
            if self.sync_bn:
                dx, dg, db, dres = B.bn_sync_backward(xv, v[:n], y.g[:n], b.weight.detach(), mean, rstd, n_glob, relu,
                                                      res is not None, zero_row=True)
            else:
                dx, dg, db, dres = B.bn_train_backward(xv, v[:n], y.g[:n], b.weight.detach(), mean, rstd, relu,
                                                       res is not None, zero_row=True)
            self._pgrad(b.weight, dg)
            self._pgrad(b.bias, db)
            x.add_grad(dx)
            if res is not None:
                res.add_grad(dres)
        self.steps.append(back)
        return y

    def _cat(self, a: _T, b: _T) -> _T:
        y = _T(torch.cat([a.v, b.v], 1), a.level)
        ca = a.v.shape[1]

        def back():
            a.add_grad(y.g[:, :ca].contiguous())
            b.add_grad(y.g[:, ca:].contiguous())
        self.steps.append(back)
        return y

    def _block(self, blk, x: _T) -> _T:
        """BasicBlock.forward (resnet_block.py:48-64)."""
        out = self._bn(self._conv(L.OP_CONV3, x, blk.conv1), blk.norm1)
        out = self._conv(L.OP_CONV3, out, blk.conv2)
        res = x
        if blk.downsample is not None:
            res = self._bn(self._conv(L.OP_LINEAR, x, blk.downsample[0]), blk.downsample[1], relu=False)
        return self._bn(out, blk.norm2, res=res, relu=True)

    def _layer(self, blocks, x: _T) -> _T:
        for blk in blocks:
            x = self._block(blk, x)
        return x

    # ------------------------------------------------------------------ forward (res16unet.py:222-295)
    def _forward(self):
        bb, sc = self.model.backbone, self.scene
        w0 = bb.conv0p1s1.kernel3().detach().contiguous()
        stem = _T(_run_stem(sc, w0, self.feats3, w0.shape[0]), 0)

        def stem_back():
            self._pgrad(bb.conv0p1s1.kernel, B.stem_weight_grad(sc, self.feats3, stem.g[:sc.n[0]], w0.shape[0]))
        self.steps.append(stem_back)
        out_p1 = self._bn(stem, bb.bn0)
        out = self._bn(self._conv(L.OP_DOWN, out_p1, bb.conv1p1s2), bb.bn1)
        out_b1p2 = self._layer(bb.block1, out)
        out = self._bn(self._conv(L.OP_DOWN, out_b1p2, bb.conv2p2s2), bb.bn2)
        out_b2p4 = self._layer(bb.block2, out)
        out = self._bn(self._conv(L.OP_DOWN, out_b2p4, bb.conv3p4s2), bb.bn3)
        out_b3p8 = self._layer(bb.block3, out)
        out = self._bn(self._conv(L.OP_DOWN, out_b3p8, bb.conv4p8s2), bb.bn4)
        out = self._layer(bb.block4, out)
        out = self._bn(self._conv(L.OP_UP, out, bb.convtr4p16s2), bb.bntr4)
        out = self._layer(bb.block5, self._cat(out, out_b3p8))
        out = self._bn(self._conv(L.OP_UP, out, bb.convtr5p8s2), bb.bntr5)
        out = self._layer(bb.block6, self._cat(out, out_b2p4))
        out = self._bn(self._conv(L.OP_UP, out, bb.convtr6p4s2), bb.bntr6)
        out = self._layer(bb.block7, self._cat(out, out_b1p2))
        out = self._bn(self._conv(L.OP_UP, out, bb.convtr7p2s2), bb.bntr7)
        out = self._layer(bb.block8, self._cat(out, out_p1))
        # lin_squeeze_head: 1x1 conv + bias (agile3d.py:43-45,179), back to the caller's row order
        head = self.model.lin_squeeze_head
        y = self._conv(L.OP_LINEAR, out, head)
        self.head_out = y
        self.orig_row = sc.table_dev(0, L.TAB_ORIGROW)[:sc.n[0]].long()
        bias = head.bias.detach().reshape(1, -1)
        n0 = sc.n[0]
        pcd = torch.empty((n0, y.v.shape[1]), dtype=torch.float32, device=y.v.device)
        pcd[self.orig_row] = y.v[:n0] + bias
        self.output = pcd

    def release(self):
        """Drop the recorded steps and activations (the backward closures and the tape refer to each other: without this the
        activations of an iteration live until Python's cycle collector runs, see DecoderTape.release)."""
        self.steps, self.relu_levels = [], []
        self.head_out = None

    # ------------------------------------------------------------------ backward
    def backward(self, d_output: torch.Tensor, on_grad=None) -> dict:
        """``d_output`` = dL/d(pcd_features) [N, 128] in the caller's row order -> gradients keyed like state_dict().
        ``on_grad(name, tensor)`` is called for every parameter as soon as its gradient is final (every backbone
        parameter is used by exactly one layer): the data-parallel all-reduce starts on the upper U-Net's gradients
        while the encoder is still being differentiated (optim.OverlappedAllReduce)."""
        head = self.model.lin_squeeze_head
        n0 = self.scene.n[0]
        g = torch.zeros((n0 + 1, d_output.shape[1]), dtype=torch.float32, device=d_output.device)
        g[:n0] = d_output.to(torch.float32)[self.orig_row]                    # internal row order + the zero row
        self.grads = {}
        self._pgrad(head.bias, B.column_sums(g[:n0]))
        self.head_out.g = g
        seen = set()
        for back in reversed(self.steps):
            back()
            if on_grad is not None and len(self.grads) != len(seen):
                for k in self.grads:
                    if k not in seen:
                        seen.add(k)
                        on_grad(k, self.grads[k])
        return self.grads
