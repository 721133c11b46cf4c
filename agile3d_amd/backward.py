"""Backward of the sparse convolutions (first part of SURVEY.md section 8 row f-2, the training path):

    conv_input_grad(scene, kind, level_in, w, dy)        dL/dx of conv / conv_tr (models/modules/common.py:125-188)
    conv_weight_grad(scene, kind, level_in, x, dy, K)    dL/dW [K, Cin, Cout]

for the four kernel-map kinds of the backbone (3^3 stride 1, 2^3 stride 2, 2^3 transposed, 1x1) -- what autograd does
inside MinkowskiEngine for ``engine.py:137-150`` (``losses.backward()``) -- plus the wrappers of the other training
kernels (BatchNorm in training mode, LayerNorm, column sums, the input conv's weight gradient).  The tapes that string
them together are ``train_backbone.py`` / ``train_decoder.py``; the iteration is ``train_step.py``.

The input gradient needs no new kernel: it is the FORWARD kernel on the transposed kernel map with transposed
weights -- for the 3^3 stride-1 map offset k of row i is row j exactly when offset 26-k of j is i, so
dx = conv3(dy, W'[k] = W[26-k]^T) on the same neighbour tables; the stride-2 conv's input gradient is the transposed
conv with W_s^T and vice versa; 1x1 is a GEMM with W^T.  The weight gradient is its own kernel (csrc/wgrad.hip).
All tensors are in the scene's internal row order (the order of the op program's activation buffers), on the GPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L

_KVOL = {L.OP_CONV3: 27, L.OP_DOWN: 8, L.OP_UP: 8, L.OP_LINEAR: 1}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_arena: dict = {}
_POISON = __import__("os").environ.get("A3D_POISON", "0") == "1"   # debugging aid, see tests/conftest.py


def _workspace(nbytes, device, tag):
    """One growing scratch allocation per (device, stream, tag): the training tapes issue hundreds of calls per iteration
    on one stream, each used to allocate (and zero) its own workspace."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, tag)
    ws = _arena.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _arena[key] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
    if _POISON:
        ws.fill_(255)   # NaN patterns: a kernel that reads scratch it did not write shows up in the results
    return ws


def _rows(t):
    """A [rows, C] operand the kernels can read in place: unit stride along the channels, any row stride (a column slice
    of a wider buffer is passed as pointer + leading dimension, never copied)."""
    return t if t.stride(1) == 1 and t.stride(0) % 4 == 0 else t.contiguous()


class StateArena:
    """The conv kernels' hand-off state (ticket, failure word, flags: a3d_conv_state_bytes() per launch) for every conv of
    a training iteration, zeroed with ONE memset when the tape starts instead of one per launch."""

    def __init__(self, device, slots):
        self.bytes = int(L.load().a3d_conv_state_bytes())
        self.buf = torch.zeros(slots * self.bytes, dtype=torch.uint8, device=device)
        self.next, self.slots = 0, slots

    def take(self):
        if self.next >= self.slots:          # more launches than planned: a fresh zeroed block
            self.buf = torch.zeros(self.slots * self.bytes, dtype=torch.uint8, device=self.buf.device)
            self.next = 0
        p = self.buf.data_ptr() + self.next * self.bytes
        self.next += 1
        return C.c_void_p(p)


def level_out(kind, level_in):
    return level_in + (1 if kind == L.OP_DOWN else -1 if kind == L.OP_UP else 0)


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[K, Cin, Cout] (ME layout) -> the MFMA fragment order the conv kernels read (a3d_pack_conv_weight)."""
    lib = L.load()
    w = w.contiguous()
    out = torch.empty(lib.a3d_conv_weight_packed_floats(w.shape[0], w.shape[1], w.shape[2]), dtype=torch.float32, device=w.device)
    L.check(lib.a3d_pack_conv_weight(_ptr(w), w.shape[0], w.shape[1], w.shape[2], _ptr(out), _stream()),
            "a3d_pack_conv_weight")
    return out


def run_conv(scene, kind, level_in, w_packed, x, cin, cout):
    """One conv of the op program on its own: x [n_in, cin] -> [n_out, cout] (no BatchNorm / ReLU / residual)."""
    lib = L.load()
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("agile3d_amd.backward runs on the GPU only (fp32 CUDA tensors)")
    lo = level_out(kind, level_in)
    n_in, n_out = scene.n[level_in], scene.n[lo]
    if x.shape != (n_in, cin):
        raise ValueError(f"input must be [{n_in}, {cin}], got {tuple(x.shape)}")
    bufs = (L.BufDesc * 2)(L.BufDesc(level_in, cin), L.BufDesc(lo, cout))
    o = L.Op()
    o.kind, o.level_in, o.cin, o.cout = kind, level_in, cin, cout
    o.in_buf, o.in_coff, o.out_buf, o.out_coff = 0, 0, 1, 0
    o.res_buf, o.res_coff, o.relu, o.kernel_volume = L.BUF_NONE, 0, 0, _KVOL[kind]
    o.w_dev, o.scale_dev, o.shift_dev = w_packed.data_ptr(), None, None
    ops = (L.Op * 1)(o)
    nbytes = lib.a3d_program_workspace_bytes(scene.handle, bufs, 2, ops, 1)
    if nbytes == 0:
        raise L.A3DError(lib.a3d_last_error().decode())
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)

    def view(i, rows, ch):
        off = lib.a3d_program_buffer_offset(scene.handle, bufs, 2, i)
        return ws[off:off + (rows + 1) * ch * 4].view(torch.float32).view(rows + 1, ch)
    view(0, n_in, cin)[:n_in].copy_(x)
    L.check(lib.a3d_program_run(scene.handle, bufs, 2, ops, 1, None, None, 0, _ptr(ws), nbytes, _stream()),
            "a3d_program_run")
    return view(1, n_out, cout)[:n_out].clone()


def conv_apply(scene, kind, level_in, w_packed, x, cin, cout, out=None, out_cols=None, zero_row=True):
    """a3d_conv_apply: one conv directly on the caller's tensors.  ``x`` [n_in + 1, ldx] with its zero row (row n_in);
    returns y [n_out + 1, cout] (row n_out zeroed) or writes ``cout`` columns into ``out[:, out_cols[0]:]``."""
    lib = L.load()
    lo = level_out(kind, level_in)
    n_in, n_out = scene.n[level_in], scene.n[lo]
    if x.shape[0] != n_in + 1 or not x.is_contiguous():
        raise ValueError(f"conv_apply: input must be a contiguous [{n_in + 1}, C] tensor carrying the zero row")
    if out is None:
        out = torch.empty((n_out + 1, cout), dtype=torch.float32, device=x.device)
        y, ldy = out, cout
    else:
        ldy = out.shape[1]
        y = out[:, out_cols[0]:] if out_cols else out
    nbytes = lib.a3d_conv_apply_workspace_bytes(scene.handle, kind, level_in, cin, cout)
    if nbytes == 0:
        raise L.A3DError(lib.a3d_last_error().decode())
    ws = _workspace(nbytes, x.device, "conv")
    L.check(lib.a3d_conv_apply(scene.handle, kind, level_in, _ptr(x), x.shape[1], cin, _ptr(w_packed), cout,
                               C.c_void_p(y.data_ptr()), ldy, int(zero_row), _ptr(ws), ws.numel(), _stream()),
            "a3d_conv_apply")
    return out


def conv_apply_acc(scene, kind, level_in, w_packed, x, cin, cout, y, acc=False, zero_row=True, state=None):
    """a3d_conv_apply_acc: y[:, :cout] (a [n_out + 1, >= cout] view, any row stride) = conv(x) (+ y when ``acc``: the
    accumulation happens in the conv kernel's epilogue).  ``x`` [n_in + 1, >= cin] view carrying the zero row."""
    lib = L.load()
    x = _rows(x)
    if y.stride(1) != 1:
        raise ValueError("conv_apply_acc: the output view must have unit channel stride")
    nbytes = lib.a3d_conv_apply_workspace_bytes(scene.handle, kind, level_in, cin, cout)
    if nbytes == 0:
        raise L.A3DError(lib.a3d_last_error().decode())
    ws = _workspace(nbytes, x.device, "conv")
    L.check(lib.a3d_conv_apply_acc(scene.handle, kind, level_in, _ptr(x), x.stride(0), cin, _ptr(w_packed), cout,
                                   _ptr(y), y.stride(0), int(zero_row), _ptr(y) if acc else None, y.stride(0) if acc else 0,
                                   state.take() if state is not None else None, _ptr(ws), ws.numel(), _stream()),
            "a3d_conv_apply_acc")
    return y


def conv_input_grad_into(scene, kind, level_in, parts, dy, cin, cout, out, acc=False, state=None):
    """dL/dx written (or, with ``acc``, added in the kernels' epilogues) into ``out`` [n_in + 1, >= cin] (a view: the
    gradient buffer of a node, possibly a column slice of a concatenation's); ``dy`` [n_out + 1, cout] with its zero row."""
    back_kind = {L.OP_CONV3: L.OP_CONV3, L.OP_DOWN: L.OP_UP, L.OP_UP: L.OP_DOWN, L.OP_LINEAR: L.OP_LINEAR}[kind]
    lo = level_out(kind, level_in)
    for c0, width, wp in parts:
        conv_apply_acc(scene, back_kind, lo, wp, dy, cout, width, out[:, c0:c0 + width], acc=acc, state=state)
    return out


def conv_bn_train_forward(scene, kind, level_in, w_packed, x, cin, cout, gamma, beta, eps, res, relu, y, running_mean,
                          running_var, momentum, state=None):
    """a3d_conv_bn_train_forward: raw = conv(x); y = relu?(BatchNorm_train(raw) (+ res)) with the batch statistics taken
    in the conv kernel's epilogue.  ``x`` [n_in + 1, >= cin] view with its zero row, ``y`` [n_out + 1, >= cout] view (row
    n_out is written as zeros), ``res`` [>= n_out, >= cout] view or None.  Returns (raw [n_out, cout], mean, rstd)."""
    lib = L.load()
    x = _rows(x)
    lo = level_out(kind, level_in)
    n_out = scene.n[lo]
    raw = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    mean = torch.empty(cout, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    nbytes = lib.a3d_conv_bn_train_workspace_bytes(scene.handle, kind, level_in, cin, cout)
    if nbytes == 0:
        raise L.A3DError(lib.a3d_last_error().decode())
    ws = _workspace(nbytes, x.device, "convbn")
    res = _rows(res) if res is not None else None
    L.check(lib.a3d_conv_bn_train_forward(scene.handle, kind, level_in, _ptr(x), x.stride(0), cin, _ptr(w_packed), cout,
                                          _ptr(raw), cout, _ptr(gamma), _ptr(beta), eps, _ptr(res),
                                          res.stride(0) if res is not None else 0, int(relu), _ptr(y), y.stride(0), 1,
                                          _ptr(mean), _ptr(rstd), _ptr(running_mean), _ptr(running_var), momentum,
                                          state.take() if state is not None else None, _ptr(ws), ws.numel(), _stream()),
            "a3d_conv_bn_train_forward")
    return raw, mean, rstd


def conv_dgrad_bn(scene, kind, level_in, part, dy, cout_fwd, out, acc, y, raw, mean, rstd, relu, state=None):
    """a3d_conv_dgrad_bn for the conv y' = conv(x; w) of kind / level_in (FORWARD op) whose input x is the output of a
    BatchNorm(+ReLU) unit (y, raw, mean, rstd): ``out`` [n_in + 1, cin] gets g = (dL/dx (+ out when ``acc``)) masked by y > 0;
    returns the fp64 sums [2, cin] (sum g, sum g xhat).  ``part`` = the single (c0 = 0, width = cin, packed) entry of
    packed_input_grad_weights, ``dy`` [n_out + 1, cout_fwd] with its zero row."""
    lib = L.load()
    back_kind = {L.OP_CONV3: L.OP_CONV3, L.OP_DOWN: L.OP_UP, L.OP_UP: L.OP_DOWN, L.OP_LINEAR: L.OP_LINEAR}[kind]
    lo = level_out(kind, level_in)
    c0, width, wp = part
    dy = _rows(dy)
    nbytes = lib.a3d_conv_bn_train_workspace_bytes(scene.handle, back_kind, lo, cout_fwd, width)
    if nbytes == 0:
        raise L.A3DError(lib.a3d_last_error().decode())
    ws = _workspace(nbytes, dy.device, "convbn")
    sums = torch.empty((2, width), dtype=torch.float64, device=dy.device)
    yy = _rows(y) if relu else None
    L.check(lib.a3d_conv_dgrad_bn(scene.handle, back_kind, lo, _ptr(dy), dy.stride(0), cout_fwd, _ptr(wp), width, _ptr(out),
                                  out.stride(0), int(acc), _ptr(yy), yy.stride(0) if yy is not None else 0, _ptr(raw),
                                  raw.stride(0), _ptr(mean), _ptr(rstd), int(relu), _ptr(sums),
                                  state.take() if state is not None else None, _ptr(ws), ws.numel(), _stream()),
            "a3d_conv_dgrad_bn")
    return sums


def bn_backward_from_sums(x, g, gamma, mean, rstd, sums, dx):
    """BatchNorm backward from the sums a3d_conv_dgrad_bn produced: g [>= n, C] (already masked by the ReLU), x = raw [n, C],
    dx [n + 1, C] (row n written as zeros).  Returns (dgamma, dbeta)."""
    lib = L.load()
    n, C_ = x.shape
    x, g = _rows(x), _rows(g)
    dgamma = torch.empty(C_, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    L.check(lib.a3d_bn_backward_apply(_ptr(x), x.stride(0), None, 0, _ptr(g), g.stride(0), n, C_, _ptr(gamma), _ptr(mean),
                                      _ptr(rstd), 0, _ptr(sums), n, _ptr(sums), _ptr(dx), dx.stride(0), None, 0, _ptr(dgamma),
                                      _ptr(dbeta), 1, _stream()), "a3d_bn_backward_apply")
    return dgamma, dbeta


def bn_train_backward_into(x, y, dy, gamma, mean, rstd, relu, dx, dres=None):
    """a3d_bn_train_backward on views: x (raw) [n, C], y / dy [>= n, C] views (any row stride), dx [n + 1, C] (row n is
    written as zeros), dres view [n + 1, C] or None.  Returns (dgamma, dbeta)."""
    lib = L.load()
    n, C_ = x.shape
    x, dy = _rows(x), _rows(dy)
    yy = _rows(y) if relu else None
    dgamma = torch.empty(C_, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = _ws(n, C_, x.device)
    L.check(lib.a3d_bn_train_backward(_ptr(x), x.stride(0), _ptr(yy), yy.stride(0) if yy is not None else 0, _ptr(dy),
                                      dy.stride(0), n, C_, _ptr(gamma), _ptr(mean), _ptr(rstd), int(relu), _ptr(dx),
                                      dx.stride(0), _ptr(dres), dres.stride(0) if dres is not None else 0, _ptr(dgamma),
                                      _ptr(dbeta), 1, _ptr(ws), ws.numel(), _stream()), "a3d_bn_train_backward")
    return dgamma, dbeta


def packed_input_grad_weights(kind, w: torch.Tensor):
    """The packed weight slices of the input-gradient conv of y = conv(x; w): [(first input channel, width, packed)].
    W' = W^T (3^3: offsets reversed); output widths the conv kernel supports (192 = 128 + 64 ...)."""
    K, cin, cout = w.shape
    wt = w.transpose(1, 2)
    if kind == L.OP_CONV3:
        wt = wt.flip(0)
    parts, c0 = [], 0
    while c0 < cin:
        rest = cin - c0
        width = rest if (rest % 128 == 0 or rest in (32, 64, 96)) else max(w_ for w_ in (128, 96, 64, 32) if w_ <= rest)
        parts.append((c0, width, pack_weight(wt[:, :, c0:c0 + width].contiguous())))
        c0 += width
    return parts


def conv_input_grad_apply(scene, kind, level_in, parts, dy, cin, cout):
    """dL/dx [n_in + 1, cin] (with its zero row) from dL/dy [n_out + 1, cout] and ``packed_input_grad_weights``."""
    back_kind = {L.OP_CONV3: L.OP_CONV3, L.OP_DOWN: L.OP_UP, L.OP_UP: L.OP_DOWN, L.OP_LINEAR: L.OP_LINEAR}[kind]
    lo = level_out(kind, level_in)
    dx = torch.empty((scene.n[level_in] + 1, cin), dtype=torch.float32, device=dy.device)
    for c0, width, wp in parts:
        conv_apply(scene, back_kind, lo, wp, dy, cout, width, out=dx, out_cols=(c0, c0 + width))
    return dx


def conv_input_grad(scene, kind, level_in, w: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """dL/dx [n_in, Cin] of y = conv(x; W) from dL/dy [n_out, Cout]; ``w`` is the forward weight [K, Cin, Cout]."""
    K, cin, cout = w.shape
    if K != _KVOL[kind]:
        raise ValueError(f"kind {kind} needs kernel volume {_KVOL[kind]}, got {K}")
    wt = w.transpose(1, 2)                               # [K, Cout, Cin]
    if kind == L.OP_CONV3:
        wt = wt.flip(0)                                  # offset k of the backward map is offset 26-k of the forward map
    back_kind = {L.OP_CONV3: L.OP_CONV3, L.OP_DOWN: L.OP_UP, L.OP_UP: L.OP_DOWN, L.OP_LINEAR: L.OP_LINEAR}[kind]
    dy = dy.contiguous()
    # the conv kernels write 32 / 64 / 96 or multiples of 128 output columns: the concatenated inputs of the decoder
    # blocks (192 = 128 + 64 channels) get their gradient in column slices
    parts, c0 = [], 0
    while c0 < cin:
        rest = cin - c0
        width = rest if (rest % 128 == 0 or rest in (32, 64, 96)) else max(w_ for w_ in (128, 96, 64, 32) if w_ <= rest)
        parts.append(run_conv(scene, back_kind, level_out(kind, level_in), pack_weight(wt[:, :, c0:c0 + width].contiguous()),
                              dy, cout, width))
        c0 += width
    return parts[0] if len(parts) == 1 else torch.cat(parts, 1)


def conv_weight_grad(scene, kind, level_in, x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """dL/dW [K, Cin, Cout] = sum over the kernel map's (input row, output row) pairs of offset k of x_in^T dy_out."""
    lib = L.load()
    if not (x.is_cuda and dy.is_cuda and x.dtype == dy.dtype == torch.float32):
        raise RuntimeError("agile3d_amd.backward runs on the GPU only (fp32 CUDA tensors)")
    lo = level_out(kind, level_in)
    cin, cout = x.shape[1], dy.shape[1]
    if x.shape[0] != scene.n[level_in] or dy.shape[0] != scene.n[lo]:
        raise ValueError("x / dy row counts do not match the scene levels")
    x, dy = _rows(x), _rows(dy)
    K = _KVOL[kind]
    dw = torch.empty((K, cin, cout), dtype=torch.float32, device=x.device)
    scene.prepare_wgrad()
    nbytes = lib.a3d_conv_wgrad_workspace_bytes(scene.handle, kind, level_in, cin, cout)
    if nbytes == 0:
        raise L.A3DError(lib.a3d_last_error().decode())
    ws = _workspace(nbytes, x.device, "wgrad")
    L.check(lib.a3d_conv_wgrad(scene.handle, kind, level_in, _ptr(x), x.stride(0), _ptr(dy), dy.stride(0), cin, cout,
                               _ptr(dw), _ptr(ws), ws.numel(), _stream()), "a3d_conv_wgrad")
    return dw


def _ws(n, C, device):
    lib = L.load()
    return _workspace(lib.a3d_bn_workspace_bytes(n, C), device, "bn")


def bn_train_forward(x, gamma, beta, eps=1e-5, res=None, relu=False, running_mean=None, running_var=None, momentum=0.1,
                     zero_row=False):
    """ME.MinkowskiBatchNorm in training mode (+ residual, + ReLU): returns (y, save_mean, save_rstd); running
    statistics, when given, are updated in place like torch's.  ``zero_row``: y gets one extra all-zero row (what a
    missing neighbour gathers when y feeds a3d_conv_apply)."""
    lib = L.load()
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("agile3d_amd.backward runs on the GPU only (fp32 CUDA tensors)")
    x = x.contiguous()
    n, C_ = x.shape
    y = torch.empty((n + 1 if zero_row else n, C_), dtype=torch.float32, device=x.device)
    mean = torch.empty(C_, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    res = res.contiguous() if res is not None else None
    ws = _ws(n, C_, x.device)
    L.check(lib.a3d_bn_train_forward(_ptr(x), C_, n, C_, _ptr(gamma), _ptr(beta), eps, _ptr(res), C_, int(relu), _ptr(y),
                                     C_, _ptr(mean), _ptr(rstd), _ptr(running_mean), _ptr(running_var), momentum,
                                     int(zero_row), _ptr(ws), ws.numel(), _stream()), "a3d_bn_train_forward")
    return y, mean, rstd


def bn_train_backward(x, y, dy, gamma, mean, rstd, relu=False, want_dres=False, zero_row=False):
    """-> (dx, dgamma, dbeta, dres or None); ``y`` (the forward output) gives the ReLU mask.  ``zero_row``: dx / dres get
    one extra all-zero row (they feed a3d_conv_apply as the input-gradient convs' dy)."""
    lib = L.load()
    x, dy = x.contiguous(), dy.contiguous()
    n, C_ = x.shape
    rows = n + 1 if zero_row else n
    dx = torch.empty((rows, C_), dtype=torch.float32, device=x.device)
    dres = torch.empty((rows, C_), dtype=torch.float32, device=x.device) if want_dres else None
    dgamma = torch.empty(C_, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = _ws(n, C_, x.device)
    L.check(lib.a3d_bn_train_backward(_ptr(x), C_, _ptr(y.contiguous()) if relu else None, C_, _ptr(dy), C_, n, C_,
                                      _ptr(gamma), _ptr(mean), _ptr(rstd), int(relu), _ptr(dx), C_, _ptr(dres), C_,
                                      _ptr(dgamma), _ptr(dbeta), int(zero_row), _ptr(ws), ws.numel(), _stream()),
            "a3d_bn_train_backward")
    return dx, dgamma, dbeta, dres


def bn_sync_forward(x, gamma, beta, eps=1e-5, res=None, relu=False, running_mean=None, running_var=None, momentum=0.1,
                    group=None, zero_row=False):
    """BatchNorm in training mode with statistics over the rows of ALL data-parallel ranks (SyncBN): the reference
    normalises over every row of the batch on one device (models/modules/common.py:20-22); with the batch's scenes
    spread over ranks the strict equivalent exchanges [2C+1] numbers per layer.  Per rank: row
    count, mean, sum of squared deviations (a3d_bn_local_stats, fp64) -> one all_gather -> Chan's parallel combination
    -> the same mean / rstd on every rank -> a3d_bn_apply.  Returns (y, mean, rstd, n_global)."""
    import torch.distributed as dist
    lib = L.load()
    x = x.contiguous()
    n, C_ = x.shape
    ws = _ws(n, C_, x.device)
    st = torch.empty(2 * C_ + 1, dtype=torch.float64, device=x.device)
    L.check(lib.a3d_bn_local_stats(_ptr(x), C_, n, C_, _ptr(st), _ptr(ws), ws.numel(), _stream()), "a3d_bn_local_stats")
    st[2 * C_] = float(n)
    from .optim import dist_all_gather
    allst = dist_all_gather(st, group)                           # [world, 2C+1]
    cnt = allst[:, 2 * C_:2 * C_ + 1]                            # [world, 1]
    n_glob = cnt.sum()
    mean = (allst[:, :C_] * cnt).sum(0) / n_glob
    m2 = (allst[:, C_:2 * C_] + cnt * (allst[:, :C_] - mean) ** 2).sum(0)
    var = m2 / n_glob
    mean_f = mean.to(torch.float32)
    rstd = (1.0 / torch.sqrt(var.to(torch.float32) + eps))
    if running_mean is not None:
        unbiased = (m2 / (n_glob - 1.0)).to(torch.float32) if float(n_glob) > 1 else var.to(torch.float32)
        running_mean.mul_(1.0 - momentum).add_(mean_f, alpha=momentum)
        running_var.mul_(1.0 - momentum).add_(unbiased, alpha=momentum)
    y = torch.empty((n + 1 if zero_row else n, C_), dtype=torch.float32, device=x.device)
    res = res.contiguous() if res is not None else None
    L.check(lib.a3d_bn_apply(_ptr(x), C_, n, C_, _ptr(gamma), _ptr(beta), _ptr(mean_f), _ptr(rstd), _ptr(res), C_,
                             int(relu), _ptr(y), C_, int(zero_row), _stream()), "a3d_bn_apply")
    return y, mean_f, rstd, int(n_glob.item())


def bn_sync_backward(x, y, dy, gamma, mean, rstd, n_global, relu=False, want_dres=False, group=None, zero_row=False):
    """Backward of ``bn_sync_forward``: sum g and sum g xhat are all-reduced (2C numbers), dx uses the global sums and
    row count; dgamma / dbeta are this rank's sums (the gradient all-reduce averages them with everything else)."""
    import torch.distributed as dist
    lib = L.load()
    x, dy = x.contiguous(), dy.contiguous()
    n, C_ = x.shape
    ws = _ws(n, C_, x.device)
    local = torch.empty(2 * C_, dtype=torch.float64, device=x.device)
    yy = y.contiguous() if relu else None
    L.check(lib.a3d_bn_backward_sums(_ptr(x), C_, _ptr(yy), C_, _ptr(dy), C_, n, C_, _ptr(mean), _ptr(rstd), int(relu),
                                     _ptr(local), _ptr(ws), ws.numel(), _stream()), "a3d_bn_backward_sums")
    from .optim import dist_all_reduce
    glob = dist_all_reduce(local.clone(), group)
    rows = n + 1 if zero_row else n
    dx = torch.empty((rows, C_), dtype=torch.float32, device=x.device)
    dres = torch.empty((rows, C_), dtype=torch.float32, device=x.device) if want_dres else None
    dgamma = torch.empty(C_, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    L.check(lib.a3d_bn_backward_apply(_ptr(x), C_, _ptr(yy), C_, _ptr(dy), C_, n, C_, _ptr(gamma), _ptr(mean), _ptr(rstd),
                                      int(relu), _ptr(glob), int(n_global), _ptr(local), _ptr(dx), C_, _ptr(dres), C_,
                                      _ptr(dgamma), _ptr(dbeta), int(zero_row), _stream()), "a3d_bn_backward_apply")
    return dx, dgamma, dbeta, dres


def column_sums(x):
    lib = L.load()
    x = x.contiguous()
    n, C_ = x.shape
    out = torch.empty(C_, dtype=torch.float32, device=x.device)
    ws = _ws(n, C_, x.device)
    L.check(lib.a3d_column_sums(_ptr(x), C_, n, C_, _ptr(out), _ptr(ws), ws.numel(), _stream()), "a3d_column_sums")
    return out


def stem_weight_grad(scene, feats3, dy, kernel_volume=125):
    """dW [K, 3, 32] of the input convolution; ``feats3`` [n0, 3] in the caller's row order, ``dy`` [n0, 32] internal."""
    lib = L.load()
    feats3, dy = feats3.contiguous(), _rows(dy)
    if feats3.shape != (scene.n[0], 3) or dy.shape[0] != scene.n[0] or dy.shape[1] < 32:
        raise ValueError("stem_weight_grad: feats3 [n0, 3], dy [n0, >= 32] expected")
    dw = torch.empty((kernel_volume, 3, 32), dtype=torch.float32, device=dy.device)
    nbytes = lib.a3d_stem_wgrad_scene_workspace_bytes(scene.handle, kernel_volume)
    if nbytes == 0:
        raise L.A3DError("stem_weight_grad: kernel volume must be 125 or 27")
    ws = _workspace(nbytes, dy.device, "stem_wgrad")
    L.check(lib.a3d_stem_wgrad(scene.handle, _ptr(feats3), _ptr(dy), dy.stride(0), kernel_volume, _ptr(dw), _ptr(ws),
                               ws.numel(), _stream()), "a3d_stem_wgrad")
    return dw


def layernorm_forward(x, gamma, beta, eps=1e-5):
    lib = L.load()
    x = x.contiguous()
    n, C_ = x.shape
    y = torch.empty_like(x)
    L.check(lib.a3d_layernorm_forward(_ptr(x), C_, n, C_, _ptr(gamma), _ptr(beta), eps, _ptr(y), C_, _stream()),
            "a3d_layernorm_forward")
    return y


def layernorm_backward(x, dy, gamma, eps=1e-5):
    """-> (dx, dgamma, dbeta) of y = LayerNorm(x) over the channels of every row."""
    lib = L.load()
    x, dy = x.contiguous(), dy.contiguous()
    n, C_ = x.shape
    dx = torch.empty_like(x)
    dgamma = torch.empty(C_, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = _ws(n, C_, x.device)
    L.check(lib.a3d_layernorm_backward(_ptr(x), C_, _ptr(dy), C_, n, C_, _ptr(gamma), eps, _ptr(dx), _ptr(dgamma),
                                       _ptr(dbeta), _ptr(ws), ws.numel(), _stream()), "a3d_layernorm_backward")
    return dx, dgamma, dbeta


def linear_weight_grad(x, dy):
    """dW [Cin, Cout] = x^T dy for [n, Cin], [n, Cout] (weights in the [in, out] layout the kernels use)."""
    lib = L.load()
    x, dy = x.contiguous(), dy.contiguous()
    n, cin = x.shape
    cout = dy.shape[1]
    nbytes = lib.a3d_linear_wgrad_workspace_bytes(n, cin, cout)
    if nbytes == 0:
        raise L.A3DError(lib.a3d_last_error().decode())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty((cin, cout), dtype=torch.float32, device=x.device)
    L.check(lib.a3d_linear_wgrad(_ptr(x), cin, _ptr(dy), cout, n, cin, cout, _ptr(dw), _ptr(ws), nbytes, _stream()),
            "a3d_linear_wgrad")
    return dw


def linear_weight_grad_into(x, dy, dw, transposed=True, accumulate=False, db=None, db_accumulate=False):
    """x^T dy written (or added) into ``dw`` -- a contiguous [Cout, Cin] block (``transposed``: nn.Linear.weight's layout, e.g. a
    row slice of an in_proj matrix) or [Cin, Cout] -- and, with ``db`` [Cout], the bias gradient (column sums of dy) from the
    same pass over dy: a3d_linear_wgrad_into."""
    lib = L.load()
    x, dy = x.contiguous(), dy.contiguous()
    n, cin = x.shape
    cout = dy.shape[1]
    want = (cout, cin) if transposed else (cin, cout)
    if tuple(dw.shape) != want or not dw.is_contiguous() or dw.dtype != torch.float32 or dy.shape[0] != n:
        raise ValueError(f"linear_weight_grad_into: dw must be a contiguous fp32 {want} block")
    if db is not None and (tuple(db.shape) != (cout,) or not db.is_contiguous() or db.dtype != torch.float32):
        raise ValueError("linear_weight_grad_into: db must be a contiguous fp32 [Cout] vector")
    nbytes = lib.a3d_linear_wgrad_into_workspace_bytes(n, cin, cout)
    if nbytes == 0:
        raise L.A3DError(lib.a3d_last_error().decode())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    L.check(lib.a3d_linear_wgrad_into(_ptr(x), cin, _ptr(dy), cout, n, cin, cout, _ptr(dw), want[1], 1 if transposed else 0,
                                      1 if accumulate else 0, _ptr(db) if db is not None else None, 1 if db_accumulate else 0,
                                      _ptr(ws), nbytes, _stream()), "a3d_linear_wgrad_into")
