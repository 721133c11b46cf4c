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
    """Activation node: value [n + 1, C] (row n = the zero row; possibly a column slice of a concatenation's buffer),
    level, gradient buffer (same shape; ``ginit`` says whether something has been written to it yet -- the first
    contribution is a plain write, later ones are added inside the producing kernel's epilogue)."""
    __slots__ = ("v", "level", "g", "ginit", "bn", "pending", "sums")

    def __init__(self, v, level):
        self.v, self.level, self.g, self.ginit = v, level, None, False
        self.bn = None        # (raw, mean, rstd, relu) of the conv + BatchNorm unit that produced this node (fused path)
        self.pending = 0      # gradient contributions still to come (consumers of the forward pass)
        self.sums = None      # fp64 [2, C]: the BatchNorm-backward sums, when the LAST contribution's conv produced them

    def grad_buffer(self):
        if self.g is None:
            self.g = torch.empty((self.v.shape[0], self.v.shape[1]), dtype=torch.float32, device=self.v.device)
        return self.g


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


_side_streams = {}


def _wgrad_stream(device):
    """The stream the weight-gradient kernels of the backward pass run on (one per device): a unit's weight gradient and its
    input gradient both depend only on d(raw), so they run side by side -- k_wgrad's waves wait for gathered rows half of
    the time (0.42-0.5 of the matrix peak), the input-gradient conv's workgroups fill in.  A3D_WGRAD_STREAM=0: one stream."""
    import os
    if os.environ.get("A3D_WGRAD_STREAM", "1") == "0":
        return None
    key = str(device)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class BackboneTape:
    """Round 5: one library call per conv + BatchNorm (+ residual)(+ ReLU) unit in the forward (the batch statistics come
    out of the conv kernel's epilogue: no statistics pass over the raw output), concatenations are column slices of one
    buffer (producers write into their slice, nothing is copied), and in the backward every fan-in is summed inside the
    producing kernel's epilogue (the input-gradient convs accumulate into the node's gradient buffer): no torch add /
    cat / copy on the tape.  ``sync_bn`` and the emulated-fp32 build (A3D_CONV_EMU) keep the layer-at-a-time BatchNorm."""

    def __init__(self, model, scene, feats3: torch.Tensor, sync_bn=None):
        import os
        if not feats3.is_cuda:
            raise RuntimeError("BackboneTape runs on the GPU only")
        self.model, self.scene = model, scene
        self.sync_bn = _sync_bn_default() if sync_bn is None else bool(sync_bn)
        self.fused_bn = not self.sync_bn and os.environ.get("A3D_CONV_EMU", "0") == "0"
        self.fuse_bwd = os.environ.get("A3D_FUSE_BN_BWD", "1") != "0"     # tests / A-B: the layer-at-a-time BatchNorm backward
        self.feats3 = feats3.to(torch.float32).contiguous()
        self.steps = []          # backward closures, in forward order
        self.wgrad_stream, self._side_used = _wgrad_stream(self.feats3.device), False
        scene.prepare_wgrad()                 # on THIS stream, before the side stream's first weight gradient
        self.relu_levels = []
        self.grads = {}
        self._names = {id(p): n for n, p in model.named_parameters()}
        self.packed = packed_weights_of(model)
        self.packed.refresh_all()
        self.state = B.StateArena(self.feats3.device, 320)    # 63 forward convs + <= 4 launches per conv in the backward
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

    def _new(self, level, C_):
        return torch.empty((self.scene.n[level] + 1, C_), dtype=torch.float32, device=self.feats3.device)

    def _conv_bn(self, kind, x: _T, conv, norm, res: _T | None = None, relu=True, out=None) -> _T:
        """conv -> BatchNorm on batch statistics (+ res)(ReLU): one unit of res16unet.py:222-295 / resnet_block.py:48-64.
        ``out``: the [n_out + 1, cout] view the result is written to (a slice of a concatenation's buffer)."""
        K, cin, cout = conv.kernel3().shape
        wf, wb = self.packed.get(conv, kind)
        sc, b = self.scene, norm.bn
        self._bns.append(b)
        lo = B.level_out(kind, x.level)
        n_in, n_out = sc.n[x.level], sc.n[lo]
        yv = out if out is not None else self._new(lo, cout)
        gamma, beta = b.weight.detach(), b.bias.detach()
        n_glob = None
        if self.fused_bn:
            raw, mean, rstd = B.conv_bn_train_forward(sc, kind, x.level, wf, x.v, cin, cout, gamma, beta, b.eps,
                                                      res.v if res is not None else None, relu, yv, b.running_mean,
                                                      b.running_var, b.momentum, state=self.state)
        else:
            rawz = self._new(lo, cout)
            B.conv_apply_acc(sc, kind, x.level, wf, x.v, cin, cout, rawz, zero_row=False, state=self.state)
            raw = rawz[:n_out]
            rv = res.v[:n_out] if res is not None else None
            if self.sync_bn:
                v, mean, rstd, n_glob = B.bn_sync_forward(raw, gamma, beta, b.eps, rv, relu, b.running_mean, b.running_var,
                                                          b.momentum, zero_row=True)
            else:
                v, mean, rstd = B.bn_train_forward(raw, gamma, beta, b.eps, rv, relu, b.running_mean, b.running_var,
                                                   b.momentum, zero_row=True)
            yv.copy_(v)
        y = _T(yv, lo)
        if self.fused_bn and self.fuse_bwd:
            y.bn = (raw, mean, rstd, relu)
        x.pending += 1
        if res is not None:
            res.pending += 1
        if relu:
            self.relu_levels.append((lo, _View(yv[:n_out])))   # forward order of the ReLUs (tests read the 0/1 masks)

        def back():
            # BatchNorm: g = dy (y > 0) -> d(raw) (with its zero row: the input-gradient conv gathers it), d(res) = g
            draw = self._new(lo, cout)
            dres = None
            if res is not None:
                if res.ginit:   # (not an assert: under python -O a second writer would silently OVERWRITE the first gradient)
                    raise RuntimeError("BackboneTape: the residual branch's gradient must be the first contribution to its node "
                                       "(a block input that is also a skip tensor or another block's residual is not a "
                                       "topology this tape's first-write rule covers)")
                res.pending -= 1
                if y.sums is None:
                    dres = res.grad_buffer()
                res.ginit = True
            if y.sums is not None:
                # the last contribution to dL/dy was an input-gradient conv that masked it (g) and summed g, g xhat in its
                # epilogue: no pass over dy / y / raw for the sums, and d(res) = g IS the buffer (no copy)
                dg, db = B.bn_backward_from_sums(raw, y.g, gamma, mean, rstd, y.sums, draw)
                if res is not None:
                    res.g = y.g
            elif self.sync_bn:
                dx, dg, db, dr = B.bn_sync_backward(raw, yv[:n_out], y.g[:n_out], gamma, mean, rstd, n_glob, relu,
                                                    res is not None, zero_row=True)
                draw.copy_(dx)
                if dres is not None:
                    dres.copy_(dr)
            else:
                dg, db = B.bn_train_backward_into(raw, yv, y.g, gamma, mean, rstd, relu, draw, dres)
            self._pgrad(b.weight, dg)
            self._pgrad(b.bias, db)
            y.g = None
            # conv: weight gradient, then the input gradient written / accumulated into the input node's buffer
            side = self.wgrad_stream
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())           # d(raw) is complete
                with torch.cuda.stream(side):
                    dw = B.conv_weight_grad(sc, kind, x.level, x.v[:n_in], draw[:n_out])
                draw.record_stream(side)                                # its memory is not handed out again before the kernel is done
                self._side_used = True
            else:
                dw = B.conv_weight_grad(sc, kind, x.level, x.v[:n_in], draw[:n_out])
            self._pgrad(conv.kernel, dw)
            x.pending -= 1
            if x.pending == 0 and x.bn is not None and len(wb) == 1:
                # this conv completes dL/dx and x is the output of a conv + BatchNorm unit: mask + BatchNorm-backward sums in
                # this launch's epilogue (a3d_conv_dgrad_bn)
                xr, xm, xs, xrelu = x.bn
                x.sums = B.conv_dgrad_bn(sc, kind, x.level, wb[0], draw, cout, x.grad_buffer(), x.ginit, x.v, xr, xm, xs, xrelu,
                                         state=self.state)
            else:
                B.conv_input_grad_into(sc, kind, x.level, wb, draw, cin, cout, x.grad_buffer(), acc=x.ginit, state=self.state)
            x.ginit = True
        self.steps.append(back)
        return y

    def _cat(self, buf, a: _T, b: _T) -> _T:
        """me.cat(a, b) (res16unet.py:257,267,277,287): both producers already wrote into their column slices of ``buf``;
        in the backward the consumers write into one gradient buffer whose slices ARE the producers' gradients."""
        y = _T(buf, a.level)
        ca = a.v.shape[1]
        a.bn = b.bn = None          # their gradients arrive as column slices of the concatenation's: no fused BatchNorm backward

        def back():
            g = y.g
            a.g, a.ginit = g[:, :ca], True                 # the up-sampled half has no other consumer
            if b.g is None:
                b.g, b.ginit = g[:, ca:], True              # the skip half: its encoder-side consumer accumulates into it later
            else:
                b.g.add_(g[:, ca:])
        self.steps.append(back)
        return y

    def _block(self, blk, x: _T, out=None) -> _T:
        """BasicBlock.forward (resnet_block.py:48-64)."""
        h = self._conv_bn(L.OP_CONV3, x, blk.conv1, blk.norm1)
        res = x
        if blk.downsample is not None:
            res = self._conv_bn(L.OP_LINEAR, x, blk.downsample[0], blk.downsample[1], relu=False)
        # conv2 -> norm2 (+ residual) -> ReLU.  Backward order inside the closure list: this unit first (its d(res) is the
        # first write into the residual node), then the projection, then conv1 -- both accumulate into x
        return self._conv_bn(L.OP_CONV3, h, blk.conv2, blk.norm2, res=res, relu=True, out=out)

    def _layer(self, blocks, x: _T, out=None) -> _T:
        for i, blk in enumerate(blocks):
            x = self._block(blk, x, out if i == len(blocks) - 1 else None)
        return x

    # ------------------------------------------------------------------ forward (res16unet.py:222-295)
    def _forward(self):
        bb, sc = self.model.backbone, self.scene
        P = bb.PLANES
        w0 = bb.conv0p1s1.kernel3().detach().contiguous()
        # the concatenation buffers [up-sampled | skip] (me.cat puts the skip LAST): the encoder's skip tensors and the
        # transposed convs' outputs are written straight into their column slices
        cat8 = self._new(0, P[7] + 32)
        cat7 = self._new(1, P[6] + P[0])
        cat6 = self._new(2, P[5] + P[1])
        cat5 = self._new(3, P[4] + P[2])
        stem = _T(_run_stem(sc, w0, self.feats3, w0.shape[0]), 0)

        def stem_back():
            self._pgrad(bb.conv0p1s1.kernel, B.stem_weight_grad(sc, self.feats3, stem.g[:sc.n[0]], w0.shape[0]))
        self.steps.append(stem_back)
        out_p1 = self._bn_only(stem, bb.bn0, cat8[:, P[7]:])
        out = self._conv_bn(L.OP_DOWN, out_p1, bb.conv1p1s2, bb.bn1)
        out_b1p2 = self._layer(bb.block1, out, cat7[:, P[6]:])
        out = self._conv_bn(L.OP_DOWN, out_b1p2, bb.conv2p2s2, bb.bn2)
        out_b2p4 = self._layer(bb.block2, out, cat6[:, P[5]:])
        out = self._conv_bn(L.OP_DOWN, out_b2p4, bb.conv3p4s2, bb.bn3)
        out_b3p8 = self._layer(bb.block3, out, cat5[:, P[4]:])
        out = self._conv_bn(L.OP_DOWN, out_b3p8, bb.conv4p8s2, bb.bn4)
        out = self._layer(bb.block4, out)
        up = self._conv_bn(L.OP_UP, out, bb.convtr4p16s2, bb.bntr4, out=cat5[:, :P[4]])
        out = self._layer(bb.block5, self._cat(cat5, up, out_b3p8))
        up = self._conv_bn(L.OP_UP, out, bb.convtr5p8s2, bb.bntr5, out=cat6[:, :P[5]])
        out = self._layer(bb.block6, self._cat(cat6, up, out_b2p4))
        up = self._conv_bn(L.OP_UP, out, bb.convtr6p4s2, bb.bntr6, out=cat7[:, :P[6]])
        out = self._layer(bb.block7, self._cat(cat7, up, out_b1p2))
        up = self._conv_bn(L.OP_UP, out, bb.convtr7p2s2, bb.bntr7, out=cat8[:, :P[7]])
        out = self._layer(bb.block8, self._cat(cat8, up, out_p1))
        # lin_squeeze_head: 1x1 conv + bias (agile3d.py:43-45,179), back to the caller's row order
        head = self.model.lin_squeeze_head
        K, cin, cout = head.kernel3().shape
        wf, wb = self.packed.get(head, L.OP_LINEAR)
        n0 = sc.n[0]
        yh = self._new(0, cout)
        B.conv_apply_acc(sc, L.OP_LINEAR, 0, wf, out.v, cin, cout, yh, state=self.state)
        self.head_out = _T(yh, 0)
        x_head = out
        x_head.pending += 1

        def head_back():
            g = self.head_out.g
            self._pgrad(head.kernel, B.conv_weight_grad(sc, L.OP_LINEAR, 0, x_head.v[:n0], g[:n0]))
            x_head.pending -= 1
            if x_head.pending == 0 and x_head.bn is not None and len(wb) == 1:
                xr, xm, xs, xrelu = x_head.bn
                x_head.sums = B.conv_dgrad_bn(sc, L.OP_LINEAR, 0, wb[0], g, cout, x_head.grad_buffer(), x_head.ginit, x_head.v,
                                              xr, xm, xs, xrelu, state=self.state)
            else:
                B.conv_input_grad_into(sc, L.OP_LINEAR, 0, wb, g, cin, cout, x_head.grad_buffer(), acc=x_head.ginit,
                                       state=self.state)
            x_head.ginit = True
        self.steps.append(head_back)
        self.orig_row = sc.table_dev(0, L.TAB_ORIGROW)[:n0].long()
        bias = head.bias.detach().reshape(1, -1)
        pcd = torch.empty((n0, cout), dtype=torch.float32, device=yh.device)
        pcd[self.orig_row] = yh[:n0] + bias
        self.output = pcd

    def _bn_only(self, x: _T, norm, out) -> _T:
        """bn0 + ReLU behind the input convolution (res16unet.py:225-227): the stem kernel has no statistics epilogue."""
        b = norm.bn
        self._bns.append(b)
        n = self.scene.n[x.level]
        xv = x.v[:n]
        gamma, beta = b.weight.detach(), b.bias.detach()
        n_glob = None
        if self.sync_bn:
            v, mean, rstd, n_glob = B.bn_sync_forward(xv, gamma, beta, b.eps, None, True, b.running_mean, b.running_var,
                                                      b.momentum, zero_row=True)
        else:
            v, mean, rstd = B.bn_train_forward(xv, gamma, beta, b.eps, None, True, b.running_mean, b.running_var, b.momentum,
                                               zero_row=True)
        out.copy_(v)
        y = _T(out, x.level)
        self.relu_levels.append((x.level, _View(out[:n])))

        def back():
            dx = x.grad_buffer()
            if self.sync_bn:
                d, dg, db, _ = B.bn_sync_backward(xv, out[:n], y.g[:n], gamma, mean, rstd, n_glob, True, False, zero_row=True)
                dx.copy_(d)
            else:
                dg, db = B.bn_train_backward_into(xv, out, y.g, gamma, mean, rstd, True, dx, None)
            x.ginit = True
            self._pgrad(b.weight, dg)
            self._pgrad(b.bias, db)
        self.steps.append(back)
        return y

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
                self._join_wgrad_stream()                               # the gradients handed over are complete on THIS stream
                for k in self.grads:
                    if k not in seen:
                        seen.add(k)
                        on_grad(k, self.grads[k])
        self._join_wgrad_stream()
        return self.grads

    def _join_wgrad_stream(self):
        if self._side_used:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self._side_used = False
