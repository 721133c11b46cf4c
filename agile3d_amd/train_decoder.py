"""Training-mode forward and backward of the click decoder (``Agile3d.forward_mask`` for one batch sample,
agile3d.py:192-339) on the HIP library -- the decoder half of SURVEY.md section 8 row f-2.

    tape = DecoderTape(model, pcd_features, pos_enc, click_idx, click_time_idx)
    tape.logits                    [3 x [N, 1+K]]: aux outputs and 'pred_masks' (last)
    grads, d_pcd = tape.backward([dL/dlogits_l ...])      gradients keyed like state_dict() + dL/d(pcd_features)

The inference path runs the decoder in a handful of fused kernels that keep nothing; training needs the intermediate
activations, so this path is the plain composition of attention_block.py -- nn.Linear = ``a3d_linear`` (the MFMA GEMM
kernels), their gradients = ``a3d_linear`` with the transposed weight + ``a3d_linear_wgrad``, LayerNorm, and the
attention / mask-head primitives of csrc/attn_train.hip with the score matrices materialised.  This module is the
reverse-mode bookkeeping (which tensor feeds which op; fan-outs are tensor adds, ReLU a mask multiply).  A parity
executor: every FLOP of consequence is in libagile3d_hip, nothing is tuned yet.  Dropout is 0 in the reference's
configuration (main.py: --dropout 0.0), so training and evaluation forward agree.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import backward as B
from . import lib as L
from .engine import time_table

H, DH = 8, 16
# A3D_TRAIN_FLASH=0: the attentions over the N points keep their [8, Lq, Lk] score matrices (attn_train.hip), the path the
# flash kernels (attn_flash.hip, default) are checked against
import os as _os
FLASH = _os.environ.get("A3D_TRAIN_FLASH", "1") != "0"


class _T:
    __slots__ = ("v", "g")

    def __init__(self, v):
        self.v, self.g = v, None

    def add_grad(self, g):
        self.g = g if self.g is None else self.g + g


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pack(w_in_out):
    cin, cout = w_in_out.shape
    return B.pack_weight(w_in_out.reshape(1, cin, cout)), cin, cout


def _linear(x, packed, bias=None):
    """x [n, cin] @ w [cin, cout] (+ bias) through a3d_linear; ``packed`` = _pack(w)."""
    lib = L.load()
    wp, cin, cout = packed
    x = x.contiguous()
    n = x.shape[0]
    y = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    L.check(lib.a3d_linear(_ptr(x), cin, None, 0, n, cin, cout, _ptr(wp), None, _ptr(bias), None, 0, 0, _ptr(y), cout,
                           None, 0, _stream()), "a3d_linear")
    return y


def _apply(P, V, Lq, Lk, Hh, dh, transposed, scale, out):
    lib = L.load()
    nbytes = lib.a3d_attn_apply_workspace_bytes(Lq, Lk, Hh, dh, transposed)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
    L.check(lib.a3d_attn_apply(_ptr(P), _ptr(V), Lq, Lk, Hh, dh, transposed, scale, _ptr(out), _ptr(ws), nbytes, _stream()),
            "a3d_attn_apply")


def _col_sums(dy):
    """bias gradient: column sums for any channel count (the kernel takes 128-column slices)."""
    n, c = dy.shape
    if c <= 384 and 768 % c == 0 and c % 32 == 0:
        return B.column_sums(dy)
    return torch.cat([B.column_sums(dy[:, i:i + 128].contiguous()) for i in range(0, c, 128)])


class DecoderTape:
    def __init__(self, model, pcd_features, pos_enc, click_idx, click_time_idx):
        if not pcd_features.is_cuda:
            raise RuntimeError("DecoderTape runs on the GPU only")
        self.model = model
        self.P = dict(model.named_parameters())
        self.steps, self.grads, self._packed = [], {}, {}
        self.relu_masks, self.attn_masks, self.args = [], [], []      # what a reference needs to follow the same branch
        self._forward(pcd_features.to(torch.float32).contiguous(), pos_enc.to(torch.float32).contiguous(), click_idx,
                      click_time_idx)

    # ------------------------------------------------------------------ primitive ops with their backward
    def _pg(self, name, g):
        g = g.reshape(self.P[name].shape)
        self.grads[name] = g if name not in self.grads else self.grads[name] + g

    def add(self, a: _T, b: _T) -> _T:
        y = _T(a.v + b.v)

        def back():
            if y.g is None:
                return
            a.add_grad(y.g)
            b.add_grad(y.g)
        self.steps.append(back)
        return y

    def lin(self, x: _T, wname, bname=None, rows=None) -> _T:
        """nn.Linear with weight [out, in] (optionally the row slice ``rows`` of an in_proj matrix)."""
        W = self.P[wname].detach()
        b = self.P[bname].detach() if bname else None
        if rows is not None:
            W, b = W[rows[0]:rows[1]], (b[rows[0]:rows[1]] if b is not None else None)
        # both orientations packed once per weight version, shared by the tapes of a batch (one tape per sample)
        from .optim import WEIGHT_EPOCH
        cache = getattr(self.model, "_a3d_packed_dec", None)
        if cache is None:
            cache = {}
            object.__setattr__(self.model, "_a3d_packed_dec", cache)
        p = self.P[wname]
        ver = (int(p._version), WEIGHT_EPOCH[0], p.data_ptr())
        hit = cache.get((wname, rows))
        if hit is None or hit[0] != ver:
            W = W.contiguous()
            hit = cache[(wname, rows)] = (ver, _pack(W.t().contiguous()), _pack(W),
                                          b.contiguous() if b is not None else None)
        _, fwd_w, bwd_w, bc = hit
        y = _T(_linear(x.v, fwd_w, bc))

        def back():
            if y.g is None:
                return
            dy = y.g.contiguous()
            x.add_grad(_linear(dy, bwd_w))                               # dy @ W
            dW = B.linear_weight_grad(x.v, dy).t()                        # [out, in]
            if rows is None:
                self._pg(wname, dW)
                if bname:
                    self._pg(bname, _col_sums(dy))
            else:                                                          # slice of the packed in_proj parameters
                full = torch.zeros_like(self.P[wname])
                full[rows[0]:rows[1]] = dW
                self._pg(wname, full)
                if bname:
                    fb = torch.zeros_like(self.P[bname])
                    fb[rows[0]:rows[1]] = _col_sums(dy)
                    self._pg(bname, fb)
        self.steps.append(back)
        return y

    def relu(self, x: _T) -> _T:
        mask = x.v > 0
        self.relu_masks.append(mask)
        y = _T(x.v * mask)

        def back():
            if y.g is not None:
                x.add_grad(y.g * mask)
        self.steps.append(back)
        return y

    def ln(self, x: _T, prefix) -> _T:
        g_, b_ = self.P[prefix + "weight"].detach(), self.P[prefix + "bias"].detach()
        y = _T(B.layernorm_forward(x.v, g_, b_))

        def back():
            if y.g is None:
                return
            dx, dg, db = B.layernorm_backward(x.v, y.g, g_)
            x.add_grad(dx)
            self._pg(prefix + "weight", dg)
            self._pg(prefix + "bias", db)
        self.steps.append(back)
        return y

    def attention(self, q: _T, k: _T, v: _T, mask=None) -> _T:
        """softmax(q k^T / sqrt(dh) + mask) v per head; q [Lq,128], k / v [Lk,128]; mask uint8 [Lq,Lk] (1 = blocked)."""
        lib = L.load()
        Lq, Lk = q.v.shape[0], k.v.shape[0]
        scale = 1.0 / (DH ** 0.5)
        dev = q.v.device
        Pm = torch.empty((H, Lq, Lk), dtype=torch.float32, device=dev)
        L.check(lib.a3d_attn_scores(_ptr(q.v), _ptr(k.v), Lq, Lk, H, DH, scale, _ptr(mask), _ptr(Pm), _stream()), "scores")
        L.check(lib.a3d_softmax_rows(_ptr(Pm), H * Lq, Lk, _stream()), "softmax")
        o = torch.empty((Lq, H * DH), dtype=torch.float32, device=dev)
        _apply(Pm, v.v, Lq, Lk, H, DH, 0, 1.0, o)
        y = _T(o)

        def back():
            if y.g is None:
                return
            do = y.g.contiguous()
            dP = torch.empty_like(Pm)
            L.check(lib.a3d_attn_scores(_ptr(do), _ptr(v.v), Lq, Lk, H, DH, 1.0, None, _ptr(dP), _stream()), "scores")
            dv = torch.empty_like(v.v)
            _apply(Pm, do, Lq, Lk, H, DH, 1, 1.0, dv)
            L.check(lib.a3d_softmax_rows_backward(_ptr(Pm), _ptr(dP), H * Lq, Lk, _stream()), "softmax_bwd")   # dP <- dS
            dq = torch.empty_like(q.v)
            _apply(dP, k.v, Lq, Lk, H, DH, 0, scale, dq)
            dk = torch.empty_like(k.v)
            _apply(dP, q.v, Lq, Lk, H, DH, 1, scale, dk)
            q.add_grad(dq)
            k.add_grad(dk)
            v.add_grad(dv)
        self.steps.append(back)
        return y

    def attention_t(self, q: _T, k: _T, v: _T) -> _T:
        """The same attention for MANY queries over FEW keys (scene-to-click: 80 k points x ~20 click queries), with the
        score matrix kept transposed, [head][key][query]: the long index is the fastest one in every kernel."""
        lib = L.load()
        Lq, Lk = q.v.shape[0], k.v.shape[0]
        scale = 1.0 / (DH ** 0.5)
        dev = q.v.device
        Pt = torch.empty((H, Lk, Lq), dtype=torch.float32, device=dev)                      # P^T[h][key][query]
        L.check(lib.a3d_attn_scores(_ptr(k.v), _ptr(q.v), Lk, Lq, H, DH, scale, None, _ptr(Pt), _stream()), "scores")
        L.check(lib.a3d_softmax_cols(_ptr(Pt), H, Lk, Lq, _stream()), "softmax_cols")         # over the keys
        o = torch.empty((Lq, H * DH), dtype=torch.float32, device=dev)
        _apply(Pt, v.v, Lk, Lq, H, DH, 1, 1.0, o)                                            # o[query] = sum_key P^T v[key]
        y = _T(o)

        def back():
            if y.g is None:
                return
            do = y.g.contiguous()
            dPt = torch.empty_like(Pt)
            L.check(lib.a3d_attn_scores(_ptr(v.v), _ptr(do), Lk, Lq, H, DH, 1.0, None, _ptr(dPt), _stream()), "scores")
            dv = torch.empty_like(v.v)
            _apply(Pt, do, Lk, Lq, H, DH, 0, 1.0, dv)                                        # dv[key] = sum_query P^T dO
            L.check(lib.a3d_softmax_cols_backward(_ptr(Pt), _ptr(dPt), H, Lk, Lq, _stream()), "softmax_cols_bwd")
            dq = torch.empty_like(q.v)
            _apply(dPt, k.v, Lk, Lq, H, DH, 1, scale, dq)                                    # dq[query] = sum_key dS^T k
            dk = torch.empty_like(k.v)
            _apply(dPt, q.v, Lk, Lq, H, DH, 0, scale, dk)                                    # dk[key] = sum_query dS^T q
            q.add_grad(dq)
            k.add_grad(dk)
            v.add_grad(dv)
        self.steps.append(back)
        return y

    # ---- the same two attentions in the flash formulation (csrc/attn_flash.hip): no [8, Lq, Lk] matrix, the softmax
    # statistics of the forward pass are what the backward pass recomputes the probabilities from
    def attention_flash_c2s(self, q: _T, k: _T, v: _T, mask=None) -> _T:
        """Few queries over the N points (click-to-scene), optional uint8 mask [Lq, Lk] (1 = blocked)."""
        lib = L.load()
        Lq, Lk = q.v.shape[0], k.v.shape[0]
        dev = q.v.device
        qs = (q.v * 0.25).contiguous()                                     # 1 / sqrt(16): exact
        kv, vv = k.v.contiguous(), v.v.contiguous()
        nbytes = lib.a3d_flash_c2s_workspace_bytes(Lq, Lk)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        o = torch.empty((Lq, H * DH), dtype=torch.float32, device=dev)
        stats = torch.empty((2, H, Lq), dtype=torch.float32, device=dev)
        L.check(lib.a3d_flash_c2s_forward(_ptr(qs), _ptr(kv), _ptr(vv), _ptr(mask), Lq, Lk, _ptr(o), _ptr(stats), _ptr(ws),
                                          nbytes, _stream()), "flash_c2s_forward")
        del ws
        y = _T(o)

        def back():
            if y.g is None:
                return
            do = y.g.contiguous()
            w2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            dqs, dk, dv = torch.empty_like(qs), torch.empty_like(kv), torch.empty_like(vv)
            L.check(lib.a3d_flash_c2s_backward(_ptr(qs), _ptr(kv), _ptr(vv), _ptr(mask), Lq, Lk, _ptr(o), _ptr(stats),
                                               _ptr(do), _ptr(dqs), _ptr(dk), _ptr(dv), _ptr(w2), nbytes, _stream()),
                    "flash_c2s_backward")
            q.add_grad(dqs * 0.25)
            k.add_grad(dk)
            v.add_grad(dv)
        self.steps.append(back)
        return y

    def attention_flash_s2c(self, q: _T, k: _T, v: _T) -> _T:
        """The N points as queries over few keys (scene-to-click), no mask."""
        lib = L.load()
        Lq, Lk = q.v.shape[0], k.v.shape[0]
        dev = q.v.device
        qs = (q.v * 0.25).contiguous()
        kv, vv = k.v.contiguous(), v.v.contiguous()
        o = torch.empty((Lq, H * DH), dtype=torch.float32, device=dev)
        stats = torch.empty((Lq, H, 2), dtype=torch.float32, device=dev)
        L.check(lib.a3d_flash_s2c_forward(_ptr(qs), _ptr(kv), _ptr(vv), Lq, Lk, _ptr(o), _ptr(stats), _stream()),
                "flash_s2c_forward")
        y = _T(o)

        def back():
            if y.g is None:
                return
            do = y.g.contiguous()
            nbytes = lib.a3d_flash_s2c_workspace_bytes(Lq, Lk)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            dqs, dk, dv = torch.empty_like(qs), torch.empty_like(kv), torch.empty_like(vv)
            L.check(lib.a3d_flash_s2c_backward(_ptr(qs), _ptr(kv), _ptr(vv), Lq, Lk, _ptr(o), _ptr(stats), _ptr(do),
                                               _ptr(dqs), _ptr(dk), _ptr(dv), _ptr(ws), nbytes, _stream()),
                    "flash_s2c_backward")
            q.add_grad(dqs * 0.25)
            k.add_grad(dk)
            v.add_grad(dv)
        self.steps.append(back)
        return y

    def mha(self, prefix, query: _T, key: _T, value: _T, mask=None) -> _T:
        """nn.MultiheadAttention (attention_block.py:25-26,88-94): in_proj slices, attention, out_proj."""
        w, b = prefix + "in_proj_weight", prefix + "in_proj_bias"
        q = self.lin(query, w, b, rows=(0, 128))
        k = self.lin(key, w, b, rows=(128, 256))
        v = self.lin(value, w, b, rows=(256, 384))
        long_queries = mask is None and q.v.shape[0] >= 1024 and q.v.shape[0] > 8 * k.v.shape[0]
        long_keys = k.v.shape[0] >= 1024 and k.v.shape[0] > 8 * q.v.shape[0]
        if FLASH and long_queries:
            a = self.attention_flash_s2c(q, k, v)
        elif FLASH and long_keys:
            a = self.attention_flash_c2s(q, k, v, mask)
        else:
            a = self.attention_t(q, k, v) if long_queries else self.attention(q, k, v, mask)
        return self.lin(a, prefix + "out_proj.weight", prefix + "out_proj.bias")

    def mask_head(self, queries: _T, src: _T, groups):
        """Agile3d.mask_module (agile3d.py:342-384): per-object max over its queries of src . MLP(LN(q))."""
        lib = L.load()
        e = self.ln(queries, "decoder_norm.")
        e = self.relu(self.lin(e, "mask_embed_head.0.weight", "mask_embed_head.0.bias"))
        E = self.lin(e, "mask_embed_head.2.weight", "mask_embed_head.2.bias")
        N, Q, G = src.v.shape[0], E.v.shape[0], len(groups)
        dev = src.v.device
        lq = torch.empty((N, Q), dtype=torch.float32, device=dev)
        L.check(lib.a3d_attn_scores(_ptr(src.v), _ptr(E.v), N, Q, 1, 128, 1.0, None, _ptr(lq), _stream()), "scores")
        qb = torch.tensor([g[0] for g in groups], dtype=torch.int32, device=dev)
        qe = torch.tensor([g[1] for g in groups], dtype=torch.int32, device=dev)
        out = torch.empty((N, G), dtype=torch.float32, device=dev)
        arg = torch.empty((N, G), dtype=torch.int32, device=dev)
        L.check(lib.a3d_group_max(_ptr(lq), N, Q, _ptr(qb), _ptr(qe), G, _ptr(out), _ptr(arg), _stream()), "group_max")
        y = _T(out)
        self.args.append(arg)

        def back():
            dlq = torch.empty_like(lq)
            L.check(lib.a3d_group_max_backward(_ptr(y.g.contiguous()), _ptr(arg), N, Q, G, _ptr(dlq), _stream()), "gm_bwd")
            dsrc = torch.empty_like(src.v)
            _apply(dlq, E.v, N, Q, 1, 128, 0, 1.0, dsrc)
            dE = torch.empty_like(E.v)
            _apply(dlq, src.v, N, Q, 1, 128, 1, 1.0, dE)
            src.add_grad(dsrc)
            E.add_grad(dE)
        self.steps.append(back)
        return y

    # ------------------------------------------------------------------ forward (agile3d.py:192-323)
    def _forward(self, pcd, pos_enc, click_idx, click_time_idx):
        dev = pcd.device
        K = len(click_idx) - 1
        fg_split = [len(click_idx[str(i)]) for i in range(1, K + 1)]
        for i, c in enumerate(fg_split):
            if c == 0:   # an empty group has no maximum: the reference fails on it too (agile3d.py:353)
                raise ValueError(f"object {i + 1} has no click (the reference fails on an empty max, agile3d.py:353)")
        fg_rows = [r for i in range(1, K + 1) for r in click_idx[str(i)]]
        fg_times = [t for i in range(1, K + 1) for t in click_time_idx[str(i)]]
        bg_rows, bg_times = list(click_idx["0"]), list(click_time_idx["0"])
        tt = time_table(128, 200).to(dev)
        n_fg, n_bgl = len(fg_rows), self.P["bg_query_feat.weight"].shape[0]
        rows = torch.tensor(fg_rows + bg_rows, dtype=torch.long, device=dev)
        fixed_pos = pos_enc[rows] + tt[torch.tensor(fg_times + bg_times, dtype=torch.long, device=dev)]
        self.pcd = _T(pcd)
        pos = _T(pos_enc)
        # queries: clicked rows of pcd_features (+ learned background queries in between), their position encodings
        bgq, bgp = self.P["bg_query_feat.weight"].detach(), self.P["bg_query_pos.weight"].detach()
        q0 = _T(torch.cat([pcd[rows[:n_fg]], bgq, pcd[rows[n_fg:]]], 0).contiguous())
        qpos = _T(torch.cat([fixed_pos[:n_fg], bgp, fixed_pos[n_fg:]], 0).contiguous())

        def q0_back():
            g = q0.g
            d = torch.zeros_like(pcd)
            d.index_add_(0, rows, torch.cat([g[:n_fg], g[n_fg + n_bgl:]], 0))   # a row clicked twice gets both
            self.pcd.add_grad(d)
            self._pg("bg_query_feat.weight", g[n_fg:n_fg + n_bgl])
            if qpos.g is not None:
                self._pg("bg_query_pos.weight", qpos.g[n_fg:n_fg + n_bgl])
        self.steps.append(q0_back)
        Q = q0.v.shape[0]
        groups = [(n_fg, Q)]                       # column 0 = background queries, then the objects
        s = 0
        for c in fg_split:
            groups.append((s, s + c))
            s += c
        src, tgt, mask = self.pcd, q0, None
        self.logits_nodes = []
        for d in range(self.model.num_decoders):
            li = 0 if self.model.shared_decoder else d
            a = self.mha(f"c2s_attention.{li}.0.multihead_attn.", self.add(tgt, qpos), self.add(src, pos), src, mask)
            tgt = self.ln(self.add(tgt, a), f"c2s_attention.{li}.0.norm.")
            qk = self.add(tgt, qpos)
            a = self.mha(f"c2c_attention.{li}.0.self_attn.", qk, qk, tgt)
            tgt = self.ln(self.add(tgt, a), f"c2c_attention.{li}.0.norm.")
            h = self.relu(self.lin(tgt, f"ffn_attention.{li}.0.linear1.weight", f"ffn_attention.{li}.0.linear1.bias"))
            f = self.lin(h, f"ffn_attention.{li}.0.linear2.weight", f"ffn_attention.{li}.0.linear2.bias")
            tgt = self.ln(self.add(tgt, f), f"ffn_attention.{li}.0.norm.")
            a = self.mha(f"s2c_attention.{li}.0.multihead_attn.", self.add(src, pos), self.add(tgt, qpos), tgt)
            src = self.ln(self.add(src, a), f"s2c_attention.{li}.0.norm.")
            out = self.mask_head(tgt, src, groups)
            self.logits_nodes.append(out)
            # attention mask of the next layer from this layer's labels (agile3d.py:362-383): not differentiated
            labels = out.v.argmax(1)
            m = torch.empty((Q, pcd.shape[0]), dtype=torch.bool, device=dev)
            for g_i, (b0, b1) in enumerate(groups):
                row = labels != g_i
                if bool(row.all()):
                    row = torch.zeros_like(row)
                m[b0:b1] = row
            mask = m.to(torch.uint8).contiguous()
            self.attn_masks.append(mask)
        self.logits = [n.v for n in self.logits_nodes]

    # ------------------------------------------------------------------ backward
    def backward(self, d_logits):
        """``d_logits``: list of dL/dlogits per decoder layer ([N, 1+K] each, None = no loss on that layer)."""
        self.grads = {}
        for node, g in zip(self.logits_nodes, d_logits):
            if g is not None:
                node.g = g.to(torch.float32).contiguous()
        for n in self.logits_nodes:
            if n.g is None:
                n.g = torch.zeros_like(n.v)
        for back in reversed(self.steps):
            back()
        return self.grads, self.pcd.g
