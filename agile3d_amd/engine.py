"""Host side of the hot path: packs the model's weights for the HIP library, describes the
Res16UNet34C topology as an op program, and implements ``forward_backbone`` /
``forward_mask`` on top of the C ABI (include/agile3d_hip.h).

Mirrors the reference's host code for this path:
  * ``models/res16unet.py:222-295``  (layer sequence, skip concatenations)
  * ``models/resnet.py:96-149`` / ``models/modules/resnet_block.py:48-64`` (BasicBlock)
  * ``models/agile3d.py:141-181``   (forward_backbone, get_pos_encs)
  * ``models/agile3d.py:183-339``   (forward_mask: per-sample loop, output dict)
PyTorch is used for device memory, streams and parameter storage only.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from . import lib as L
from .sparse import SparseTensor

BN_EPS = 1e-5


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Scene:
    """Owner of one ``a3d_scene`` handle and its device workspace."""

    def __init__(self, coords: torch.Tensor):
        lib = L.load()
        assert coords.is_cuda and coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
        self.coords = coords.contiguous()
        n = self.coords.shape[0]
        nbytes = lib.a3d_scene_workspace_bytes(n)
        if nbytes == 0:
            raise L.A3DError(f"cannot build a scene of {n} voxels")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=coords.device)
        h = C.c_void_p()
        L.check(lib.a3d_scene_create(_ptr(self.coords), n, _ptr(self.workspace), nbytes, _stream(), C.byref(h)),
                "a3d_scene_create")
        self.handle = h
        self.n = [int(lib.a3d_scene_level_size(h, i)) for i in range(L.A3D_NUM_LEVELS)]
        starts = (C.c_int64 * 1024)()
        nb = lib.a3d_scene_batch_ranges(h, starts, 1024)
        ends = [int(starts[i + 1]) for i in range(nb - 1)] + [n]
        self.batch_ranges = [(int(starts[i]), ends[i]) for i in range(nb)]
        dims = (C.c_int * 3)()
        # level-0 lookups go through a dense voxel grid when the batch's bounding box is small, else a hash table
        self.grid_dims = tuple(dims) if lib.a3d_scene_grid_dims(h, dims) == 1 else None

    def prepare_wgrad(self):
        """The weight-gradient work lists of this scene (a3d_scene_build_wgrad_lists), requested on the current stream at the
        first call (no synchronisation: the first weight gradient waits for their lengths); a3d_conv_wgrad refuses a scene
        without them."""
        if getattr(self, "_wgrad_lists", None) is None:
            lib = L.load()
            nbytes = lib.a3d_scene_wgrad_lists_bytes(self.handle)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.workspace.device)
            L.check(lib.a3d_scene_build_wgrad_lists(self.handle, _ptr(ws), nbytes, _stream()), "a3d_scene_build_wgrad_lists")
            self._wgrad_lists = ws

    def table(self, level: int, which: int) -> np.ndarray:
        """Copy one scene table to the host (tests / debugging)."""
        lib = L.load()
        p, cnt = C.c_void_p(), C.c_int64()
        L.check(lib.a3d_scene_table(self.handle, level, which, C.byref(p), C.byref(cnt)), "a3d_scene_table")
        dt = np.uint32 if which in (L.TAB_GMASK27, L.TAB_GMASKDOWN, L.TAB_GMASKUP) else np.int32
        out = np.empty(cnt.value, dtype=dt)
        L.check(lib.a3d_memcpy_d2h(out.ctypes.data_as(C.c_void_p), p, out.nbytes, _stream()), "a3d_memcpy_d2h")
        return out

    def table_dev(self, level: int, which: int) -> torch.Tensor:
        """One scene table as an int32 tensor VIEW of the scene's device workspace (no copy, no synchronisation; valid as
        long as this Scene lives)."""
        lib = L.load()
        p, cnt = C.c_void_p(), C.c_int64()
        L.check(lib.a3d_scene_table(self.handle, level, which, C.byref(p), C.byref(cnt)), "a3d_scene_table")
        off = p.value - self.workspace.data_ptr()
        assert 0 <= off and off + 4 * cnt.value <= self.workspace.numel() and off % 4 == 0
        return self.workspace[off:off + 4 * cnt.value].view(torch.int32)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                L.load().a3d_scene_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _Pool:
    """Activation buffers of the op program; buffers are recycled once their last reader ran."""

    def __init__(self):
        self.descs = []
        self.free = {}

    def get(self, level, ch):
        lst = self.free.get((level, ch))
        if lst:
            return lst.pop()
        self.descs.append((level, ch))
        return len(self.descs) - 1

    def put(self, i):
        self.free.setdefault(self.descs[i], []).append(i)


class BackboneProgram:
    """Res16UNet34C + lin_squeeze_head as a list of ``a3d_op`` (built once per weight version)."""

    def __init__(self, model, device, fuse_proj=None):
        lib = L.load()
        self.keep = []          # tensors referenced by raw pointers
        self.ops = []
        self.pool = _Pool()
        self.fm_bufs = []       # buffer ids of the 5 feature maps (res16unet.py:250-290)
        bb = model.backbone
        dev = device

        # the residual projections (BasicBlock.downsample) run inside the block's second conv (a3d_op.proj_*): not with the
        # emulated-fp32 conv products (their weights are packed as bf16 planes), and A3D_FUSE_PROJ=0 keeps the separate
        # 1x1 launches (tests compare the two)
        fuse_proj = (os.environ.get("A3D_FUSE_PROJ", "1") != "0" and os.environ.get("A3D_CONV_EMU", "0") == "0"
                     if fuse_proj is None else fuse_proj)
        self.fused_projections = 0

        def pack(conv, w=None):
            w = (conv.kernel3().detach() if w is None else w).to(dev, torch.float32).contiguous()
            K, cin, cout = w.shape
            out = torch.empty(lib.a3d_conv_weight_packed_floats(K, cin, cout), dtype=torch.float32, device=dev)
            L.check(lib.a3d_pack_conv_weight(_ptr(w), K, cin, cout, _ptr(out), _stream()), "a3d_pack_conv_weight")
            self.keep.append(out)
            return out

        def fold(bn):
            b = bn.bn
            scale = (b.weight.detach() / torch.sqrt(b.running_var.detach() + b.eps)).to(dev, torch.float32)
            shift = (b.bias.detach() - b.running_mean.detach() * scale.to(b.bias.device)).to(dev, torch.float32)
            scale, shift = scale.contiguous(), shift.contiguous()
            self.keep += [scale, shift]
            return scale, shift

        def op(kind, level_in, cin, cout, src, dst, res, relu, kvol, w, scale, shift, proj=None):
            o = L.Op()
            o.kind, o.level_in, o.cin, o.cout = kind, level_in, cin, cout
            o.in_buf, o.in_coff = src
            o.out_buf, o.out_coff = dst
            o.res_buf, o.res_coff = res if res is not None else (L.BUF_NONE, 0)
            o.relu, o.kernel_volume = int(relu), kvol
            o.w_dev = w.data_ptr()
            o.scale_dev = scale.data_ptr() if scale is not None else None
            o.shift_dev = shift.data_ptr() if shift is not None else None
            if proj is not None:
                (o.proj_buf, o.proj_coff), o.proj_cin = proj[0], proj[1]
            self.ops.append(o)

        def basic_block(blk, level, src, cin, cout, dst):
            """BasicBlock.forward (resnet_block.py:48-64); src/dst = (buffer, column offset)."""
            tmp = self.pool.get(level, cout)
            s1, h1 = fold(blk.norm1)
            op(L.OP_CONV3, level, cin, cout, src, (tmp, 0), None, True, 27, pack(blk.conv1), s1, h1)
            res, proj = src, None
            if blk.downsample is not None and fuse_proj and cin % 32 == 0:
                # out = relu(s2 conv2(h) + h2 + sp (x Wp) + hp): both scales into the weights, one accumulator, one launch.
                # (the kernel's stage width at these shapes is 32 or 64 channels; the 32 -> 64 block of level 2 runs its
                # 64 -> 64 conv with 32-channel stages so that its 32-channel projection is a whole stage: all seven fused)
                s2, h2 = fold(blk.norm2)
                sp, hp = fold(blk.downsample[1])
                w2 = blk.conv2.kernel3().detach().to(dev, torch.float32) * s2[None, None, :]
                wp = blk.downsample[0].kernel3().detach().to(dev, torch.float32) * sp[None, None, :]
                both = torch.cat([pack(None, w2), pack(None, wp)]).contiguous()
                shift = (h2 + hp).contiguous()
                self.keep += [both, shift]
                op(L.OP_CONV3, level, cout, cout, (tmp, 0), dst, None, True, 27, both, None, shift, proj=(src, cin))
                self.fused_projections += 1
                self.pool.put(tmp)
                return
            if blk.downsample is not None:
                proj = self.pool.get(level, cout)
                sp, hp = fold(blk.downsample[1])
                op(L.OP_LINEAR, level, cin, cout, src, (proj, 0), None, False, 1, pack(blk.downsample[0]), sp, hp)
                res = (proj, 0)
            s2, h2 = fold(blk.norm2)
            op(L.OP_CONV3, level, cout, cout, (tmp, 0), dst, res, True, 27, pack(blk.conv2), s2, h2)
            self.pool.put(tmp)
            if proj is not None:
                self.pool.put(proj)

        def layer(blocks, level, src, cin, cout, dst, free_src):
            """_make_layer's Sequential of BasicBlocks; the last block writes to dst."""
            cur, cur_c, owned = src, cin, free_src
            n = len(blocks)
            for i, blk in enumerate(blocks):
                if i == n - 1:
                    out = dst
                else:
                    out = (self.pool.get(level, cout), 0)
                basic_block(blk, level, cur, cur_c, cout, out)
                if owned:
                    self.pool.put(cur[0])
                cur, cur_c, owned = out, cout, (i != n - 1)

        P = bb.PLANES
        # concat buffers: [upsampled | skip] -- me.cat(out, skip) puts the skip LAST (res16unet.py:257)
        cat8 = self.pool.get(0, P[7] + 32)
        cat7 = self.pool.get(1, P[6] + P[0])
        cat6 = self.pool.get(2, P[5] + P[1])
        cat5 = self.pool.get(3, P[4] + P[2])
        s, h = fold(bb.bn0)
        w0 = bb.conv0p1s1.kernel3().detach().to(dev, torch.float32).contiguous()
        self.keep.append(w0)
        op(L.OP_STEM, 0, 3, 32, (L.BUF_NONE, 0), (cat8, P[7]), None, True, w0.shape[0], w0, s, h)

        def down(conv, bn, level, src, c):
            dst = self.pool.get(level + 1, c)
            sc, sh = fold(bn)
            op(L.OP_DOWN, level, c, c, src, (dst, 0), None, True, 8, pack(conv), sc, sh)
            return dst

        x = down(bb.conv1p1s2, bb.bn1, 0, (cat8, P[7]), 32)
        layer(bb.block1, 1, (x, 0), 32, P[0], (cat7, P[6]), True)
        x = down(bb.conv2p2s2, bb.bn2, 1, (cat7, P[6]), P[0])
        layer(bb.block2, 2, (x, 0), P[0], P[1], (cat6, P[5]), True)
        x = down(bb.conv3p4s2, bb.bn3, 2, (cat6, P[5]), P[1])
        layer(bb.block3, 3, (x, 0), P[1], P[2], (cat5, P[4]), True)
        x = down(bb.conv4p8s2, bb.bn4, 3, (cat5, P[4]), P[2])
        f0 = self.pool.get(4, P[3])
        layer(bb.block4, 4, (x, 0), P[2], P[3], (f0, 0), True)

        def up(conv, bn, level_in, src, cin, dst, cout):
            sc, sh = fold(bn)
            op(L.OP_UP, level_in, cin, cout, src, dst, None, True, 8, pack(conv), sc, sh)

        up(bb.convtr4p16s2, bb.bntr4, 4, (f0, 0), P[3], (cat5, 0), P[4])
        f1 = self.pool.get(3, P[4])
        layer(bb.block5, 3, (cat5, 0), P[4] + P[2], P[4], (f1, 0), False)
        up(bb.convtr5p8s2, bb.bntr5, 3, (f1, 0), P[4], (cat6, 0), P[5])
        f2 = self.pool.get(2, P[5])
        layer(bb.block6, 2, (cat6, 0), P[5] + P[1], P[5], (f2, 0), False)
        up(bb.convtr6p4s2, bb.bntr6, 2, (f2, 0), P[5], (cat7, 0), P[6])
        f3 = self.pool.get(1, P[6])
        layer(bb.block7, 1, (cat7, 0), P[6] + P[0], P[6], (f3, 0), False)
        up(bb.convtr7p2s2, bb.bntr7, 1, (f3, 0), P[6], (cat8, 0), P[7])
        f4 = self.pool.get(0, P[7])
        layer(bb.block8, 0, (cat8, 0), P[7] + 32, P[7], (f4, 0), False)
        self.fm_bufs = [f0, f1, f2, f3, f4]
        # lin_squeeze_head: 1x1 conv + bias into the caller-ordered output (agile3d.py:179)
        head = model.lin_squeeze_head
        hb = head.bias.detach().reshape(-1).to(dev, torch.float32).contiguous()
        self.keep.append(hb)
        hw = pack(head)
        last = self.ops[-1]                   # block8's last conv: its workgroups hold complete 96-column output rows
        if (os.environ.get("A3D_FUSE_HEAD", "0") == "1" and last.kind == L.OP_CONV3 and last.out_buf == f4
                and last.proj_cin == 0):
            # lin_squeeze_head as a second GEMM in that conv's epilogue (SURVEY 2.1's second fusion): the [N, 96] rows are
            # not read back, one launch less.  The library runs the head as its own 1x1 launch where no fused build exists.
            # OPT-IN (A3D_FUSE_HEAD=1): built and measured in round 5 -- the fused launch takes 2 650-2 760 us on the 16-scene
            # batch against 2 190 + 336 for conv + k_dense (four workgroups per CU each stream the 48 KB head weights through
            # the CU's L1 for every 64-row tile), 640-643 vs 643-646 scenes/s; one scene: 208 vs 172 + 33 us
            # (profiles/r05_experiments.txt).  The separate launch stays the default.
            last.head_w_dev, last.head_bias_dev, last.head_cout = hw.data_ptr(), hb.data_ptr(), model.mask_dim
            self.fused_head = True
        else:
            op(L.OP_LINEAR, 0, P[7], model.mask_dim, (f4, 0), (L.BUF_EXT_OUT, 0), None, False, 1, hw, None, hb)
            self.fused_head = False

        self.n_ops = len(self.ops)
        self.ops_arr = (L.Op * self.n_ops)(*self.ops)
        self.n_bufs = len(self.pool.descs)
        self.bufs_arr = (L.BufDesc * self.n_bufs)(*[L.BufDesc(lv, ch) for lv, ch in self.pool.descs])

    def run(self, scene: Scene, feats: torch.Tensor, out: torch.Tensor):
        lib = L.load()
        nbytes = lib.a3d_program_workspace_bytes(scene.handle, self.bufs_arr, self.n_bufs, self.ops_arr, self.n_ops)
        if nbytes == 0:
            raise L.A3DError("a3d_program_workspace_bytes: " + lib.a3d_last_error().decode())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=feats.device)
        L.check(lib.a3d_program_run(scene.handle, self.bufs_arr, self.n_bufs, self.ops_arr, self.n_ops,
                                    _ptr(feats), _ptr(out), out.shape[1], _ptr(ws), nbytes, _stream()),
                "a3d_program_run")
        return ws

    def feature_map(self, scene: Scene, ws: torch.Tensor, i: int) -> torch.Tensor:
        lib = L.load()
        b = self.fm_bufs[i]
        off = lib.a3d_program_buffer_offset(scene.handle, self.bufs_arr, self.n_bufs, b)
        level, ch = self.pool.descs[b]
        n = scene.n[level]
        return ws[off:off + n * ch * 4].view(torch.float32).view(n, ch)


def time_table(d_model=128, length=200):
    """PositionalEncoding1D (position_embedding.py:210-225)."""
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


class DecoderPack:
    """``a3d_decoder_weights`` for the model's current parameters."""

    def __init__(self, model, device):
        lib = L.load()
        self.keep = []
        dev = device
        W = L.DecoderWeights()
        n_layers = model.num_decoders
        if n_layers > L.A3D_MAX_DEC_LAYERS:
            raise L.A3DError("too many decoder layers")
        W.n_layers = n_layers
        W.n_bg_queries = model.num_bg_queries
        W.dim_ff = model.args.dim_feedforward

        def dv(t):
            t = t.detach().to(dev, torch.float32).contiguous()
            self.keep.append(t)
            return t.data_ptr()

        def tr(t):   # query-side matrices stay in torch layout [out][in]
            return dv(t)

        def packed(wt_in_out):  # [in][out] -> MFMA fragment order
            w = wt_in_out.detach().to(dev, torch.float32).contiguous().unsqueeze(0)
            out = torch.empty_like(w)
            L.check(lib.a3d_pack_conv_weight(_ptr(w), 1, w.shape[1], w.shape[2], _ptr(out), _stream()),
                    "a3d_pack_conv_weight")
            self.keep.append(out)
            return out.data_ptr()

        d = model.mask_dim
        for l in range(n_layers):
            li = 0 if model.shared_decoder else l
            c2s = model.c2s_attention[li][0]
            c2c = model.c2c_attention[li][0]
            ffn = model.ffn_attention[li][0]
            s2c = model.s2c_attention[li][0]
            lw = W.layers[l]
            a = c2s.multihead_attn
            lw.c2s_in_w, lw.c2s_in_b = tr(a.in_proj_weight), dv(a.in_proj_bias)
            lw.c2s_out_w, lw.c2s_out_b = tr(a.out_proj.weight), dv(a.out_proj.bias)
            lw.c2s_norm_w, lw.c2s_norm_b = dv(c2s.norm.weight), dv(c2s.norm.bias)
            lw.c2s_wk_packed = packed(a.in_proj_weight[d:2 * d].t())
            lw.c2s_wv_packed = packed(a.in_proj_weight[2 * d:].t())
            a = c2c.self_attn
            lw.c2c_in_w, lw.c2c_in_b = tr(a.in_proj_weight), dv(a.in_proj_bias)
            lw.c2c_out_w, lw.c2c_out_b = tr(a.out_proj.weight), dv(a.out_proj.bias)
            lw.c2c_norm_w, lw.c2c_norm_b = dv(c2c.norm.weight), dv(c2c.norm.bias)
            lw.ffn_w1, lw.ffn_b1 = tr(ffn.linear1.weight), dv(ffn.linear1.bias)
            lw.ffn_w2, lw.ffn_b2 = tr(ffn.linear2.weight), dv(ffn.linear2.bias)
            lw.ffn_norm_w, lw.ffn_norm_b = dv(ffn.norm.weight), dv(ffn.norm.bias)
            a = s2c.multihead_attn
            lw.s2c_in_w, lw.s2c_in_b = tr(a.in_proj_weight), dv(a.in_proj_bias)
            lw.s2c_out_w, lw.s2c_out_b = tr(a.out_proj.weight), dv(a.out_proj.bias)
            lw.s2c_norm_w, lw.s2c_norm_b = dv(s2c.norm.weight), dv(s2c.norm.bias)
            lw.s2c_wq_packed = packed(a.in_proj_weight[:d].t())
            lw.s2c_wo_packed = packed(a.out_proj.weight.t())
        W.decoder_norm_w, W.decoder_norm_b = dv(model.decoder_norm.weight), dv(model.decoder_norm.bias)
        m0, m2 = model.mask_embed_head[0], model.mask_embed_head[2]
        W.mask_w0, W.mask_b0 = tr(m0.weight), dv(m0.bias)
        W.mask_w2, W.mask_b2 = tr(m2.weight), dv(m2.bias)
        W.bg_query_feat, W.bg_query_pos = dv(model.bg_query_feat.weight), dv(model.bg_query_pos.weight)
        W.gauss_B = dv(model.pos_enc.gauss_B)
        W.time_table = dv(time_table(d, 200))
        # the query side's matrices once more, in the order the single-block layer kernel's waves read them
        for l in range(n_layers):
            qp = torch.empty(lib.a3d_decoder_query_pack_floats(W.dim_ff), dtype=torch.float32, device=dev)
            L.check(lib.a3d_decoder_pack_query_weights(C.byref(W), l, _ptr(qp), _stream()), "a3d_decoder_pack_query_weights")
            self.keep.append(qp)
            W.layers[l].query_pack = qp.data_ptr()
        mp = torch.empty(lib.a3d_decoder_mask_pack_floats(), dtype=torch.float32, device=dev)
        L.check(lib.a3d_decoder_pack_query_weights(C.byref(W), -1, _ptr(mp), _stream()), "a3d_decoder_pack_query_weights")
        self.keep.append(mp)
        W.mask_pack = mp.data_ptr()
        self.W = W
        self.n_layers = n_layers
        self.gauss_B_ptr = W.gauss_B


_KV_MB = None


def _kv_cache_mb():
    """Cap of the per-scene key / value cache of forward_mask in MB (A3D_KV_CACHE_MB, default 4096; 0 = off)."""
    global _KV_MB
    if _KV_MB is None:
        import os
        _KV_MB = int(os.environ.get("A3D_KV_CACHE_MB", "4096"))
    return _KV_MB


class _SceneState:
    """Everything forward_mask needs from forward_backbone; rides on the returned pcd_features."""

    def __init__(self):
        self.scene = None
        self.ws = None
        self.ranges = None
        self.posenc = None      # list[Tensor [n_b,128]]
        self.minmax = None      # list[Tensor [6]]
        self.engine_id = None
        self.train = False      # produced by the training-mode forward (no inference workspace, aux placeholders)
        # per-scene cache of the first decoder layer's click-to-scene keys / values (click-independent): allocated and
        # filled by the SECOND forward_mask on this backbone output, read by every later one
        self.mask_calls = 0
        self.kv0 = None         # list[Tensor [3, n_b, 128]]: keys, values, scene-to-click queries of the first layer
        self.kv0_version = None


class Engine:
    def __init__(self, model, device):
        L.load()
        self.model = model
        self.device = device
        self._stale = True
        self._stale_dec = True
        self._version = None
        self._version_dec = None
        self._dec_tensors = None
        self._all_tensors = None
        self.program = None
        self.decoder = None

    def mark_stale(self):
        self._stale = True
        self._stale_dec = True
        pw = getattr(self.model, "_a3d_packed_train", None)      # the training tapes' packed conv weights
        if pw is not None:
            pw.invalidate()

    def _weights_version(self, decoder_only=False):
        if self._dec_tensors is None:
            sd = self.model.state_dict(keep_vars=True)
            self._all_tensors = list(sd.values())
            self._dec_tensors = [v for k, v in sd.items() if not k.startswith("backbone.") and not k.startswith("lin_squeeze_head.")]
        return sum(int(t._version) for t in (self._dec_tensors if decoder_only else self._all_tensors))

    def refresh_weights_if_stale(self, check_versions=False):
        """Packed weights of the inference path: the folded/packed backbone program and the decoder pack.  They are
        rebuilt when marked stale or (``check_versions``) when a parameter tensor was written since (an optimiser step)."""
        if not self._stale and check_versions:
            if self._weights_version() != self._version:
                self._stale = self._stale_dec = True
        if self._stale:
            with torch.no_grad():
                self.program = BackboneProgram(self.model, self.device)
            self._version = self._weights_version()
            self._stale = False
        self.refresh_decoder_if_stale()

    def refresh_decoder_if_stale(self, check_versions=False):
        """The decoder's packed weights only: what forward_mask needs (the no-grad click rounds of a training
        iteration, engine.py:82-115, must not re-fold and re-pack the whole backbone)."""
        if not self._stale_dec and check_versions and self._weights_version(True) != self._version_dec:
            self._stale_dec = True
        if self._stale_dec or self.decoder is None:
            with torch.no_grad():
                self.decoder = DecoderPack(self.model, self.device)
            self._version_dec = self._weights_version(True)
            self._stale_dec = False
            self._dec_epoch = getattr(self, "_dec_epoch", 0) + 1     # what a scene's cached keys / values were made with

    # ---------------------------------------------------------------- forward_backbone
    def forward_backbone(self, x, raw_coordinates=None):
        lib = L.load()
        self.refresh_weights_if_stale(check_versions=True)
        if not isinstance(x, SparseTensor):
            x = SparseTensor(features=x.F, coordinates=x.C, device=self.device)
        if x.device != self.device:
            x = SparseTensor(features=x.F, coordinates=x.C, device=self.device)
        if raw_coordinates is None:
            raise ValueError("forward_backbone needs raw_coordinates (agile3d.py:163)")
        raw = raw_coordinates.to(self.device, torch.float32).contiguous()
        n = len(x)
        if raw.shape != (n, 3):
            raise ValueError("raw_coordinates must be [N,3]")
        with torch.no_grad():
            st = _SceneState()
            st.engine_id = id(self)
            st.scene = Scene(x.C)
            out = torch.empty((n, self.model.mask_dim), dtype=torch.float32, device=self.device)
            st.ws = self.program.run(st.scene, x.F, out)
            st.ranges = st.scene.batch_ranges      # from the scene build's single read-back, no torch kernels
            st.posenc, st.minmax = self._posenc_batch(raw, st.ranges)
        pcd_features = SparseTensor(features=out, coordinates=x.C)
        pcd_features._a3d = st
        coordinates = SparseTensor(features=raw, coordinates=x.C)
        aux = _AuxList(self.program, st)
        pos_encodings_pcd = [[[None] * len(st.ranges)] for _ in range(4)] + [[list(st.posenc)]]
        return pcd_features, aux, coordinates, pos_encodings_pcd

    def forward_backbone_train(self, x, raw_coordinates=None):
        """``forward_backbone`` of a model in training mode (engine.py:53 of the reference): BatchNorm on the statistics
        of this batch (running statistics updated), every activation kept for the backward pass (BackboneTape), the
        result tied into torch.autograd -- ``pcd_features.F`` depends on every backbone parameter."""
        from .autograd import BackboneFn, _Holder
        from .train_backbone import BackboneTape
        lib = L.load()
        self.refresh_decoder_if_stale(check_versions=True)      # gauss_B for the position encodings
        if not isinstance(x, SparseTensor) or x.device != self.device:
            x = SparseTensor(features=x.F, coordinates=x.C, device=self.device)
        if raw_coordinates is None:
            raise ValueError("forward_backbone needs raw_coordinates (agile3d.py:163)")
        raw = raw_coordinates.to(self.device, torch.float32).contiguous()
        n = len(x)
        if raw.shape != (n, 3):
            raise ValueError("raw_coordinates must be [N,3]")
        st = _SceneState()
        st.engine_id = id(self)
        st.train = True
        with torch.no_grad():
            from .train_backbone import packed_weights_of
            packed_weights_of(self.model).refresh_all()      # scene-independent: queued before the scene build's host round trip
            st.scene = Scene(x.C)
            tape = BackboneTape(self.model, st.scene, x.F.detach())
            st.ranges = st.scene.batch_ranges
            st.posenc, st.minmax = self._posenc_batch(raw, st.ranges)
        named = [(k, p) for k, p in self.model.named_parameters()
                 if k.startswith("backbone.") or k.startswith("lin_squeeze_head.")]
        holder = _Holder(tape=tape, names=[k for k, _ in named])
        out = BackboneFn.apply(holder, *[p for _, p in named]) if torch.is_grad_enabled() else tape.output
        pcd_features = SparseTensor(features=out, coordinates=x.C)
        pcd_features._a3d = st
        coordinates = SparseTensor(features=raw, coordinates=x.C)
        pos_encodings_pcd = [[[None] * len(st.ranges)] for _ in range(4)] + [[list(st.posenc)]]
        return pcd_features, [None] * 5, coordinates, pos_encodings_pcd

    def forward_mask_train(self, pcd_features, aux, coordinates, pos_encodings_pcd, click_idx=None, click_time_idx=None):
        """``forward_mask`` of a model in training mode (engine.py:120-122): one DecoderTape per batch sample
        (agile3d.py:192 loops over the samples), logits tied into torch.autograd."""
        from .autograd import DecoderFn, _Holder
        from .train_decoder import DecoderTape
        st = getattr(pcd_features, "_a3d", None)
        if st is None or st.engine_id != id(self):
            raise RuntimeError("forward_mask needs the objects returned by this model's forward_backbone")
        if click_idx is None or click_time_idx is None:
            raise ValueError("click_idx and click_time_idx are required")
        named = [(k, p) for k, p in self.model.named_parameters()
                 if not (k.startswith("backbone.") or k.startswith("lin_squeeze_head."))]
        names, params = [k for k, _ in named], [p for _, p in named]
        n_layers = self.model.num_decoders
        preds = [[] for _ in range(n_layers)]
        for b, (s, e) in enumerate(st.ranges):
            rows = pcd_features.F[s:e]
            with torch.no_grad():
                tape = DecoderTape(self.model, rows.detach(), st.posenc[b], click_idx[b], click_time_idx[b])
            if torch.is_grad_enabled():
                logits = DecoderFn.apply(_Holder(tape=tape, names=names), rows, *params)
            else:
                logits = tape.logits
            for l in range(n_layers):
                preds[l].append(logits[l])
        out = {"pred_masks": preds[-1], "backbone_features": pcd_features}
        if self.model.aux:
            out["aux_outputs"] = [{"pred_masks": p} for p in preds[:-1]]
        return out

    def decoder_inputs(self, feats128: torch.Tensor, raw_xyz: torch.Tensor):
        """Single-sample decoder inputs from an explicit [N,128] feature matrix (what
        forward_backbone would have produced) -- used by the golden-vector parity tests, which
        hold the reference's decoder inputs directly."""
        lib = L.load()
        self.refresh_decoder_if_stale(check_versions=True)
        feats = feats128.to(self.device, torch.float32).contiguous()
        raw = raw_xyz.to(self.device, torch.float32).contiguous()
        n = feats.shape[0]
        C4 = torch.zeros((n, 4), dtype=torch.int32, device=self.device)
        st = _SceneState()
        st.engine_id = id(self)
        st.ranges = [(0, n)]
        with torch.no_grad():
            tmp = torch.empty(256 * 6 * 4, dtype=torch.uint8, device=self.device)
            pe = torch.empty((n, 128), dtype=torch.float32, device=self.device)
            mm = torch.empty(6, dtype=torch.float32, device=self.device)
            L.check(lib.a3d_posenc_fourier(_ptr(raw), n, self.decoder.gauss_B_ptr, _ptr(mm), _ptr(pe), _ptr(tmp),
                                           tmp.numel(), _stream()), "a3d_posenc_fourier")
        st.posenc, st.minmax = [pe], [mm]
        pcd = SparseTensor(features=feats, coordinates=C4)
        pcd._a3d = st
        coordinates = SparseTensor(features=raw, coordinates=C4)
        return pcd, None, coordinates, [[[None]] for _ in range(4)] + [[[pe]]]

    def decoder_inputs_batch(self, feats128: torch.Tensor, raw_xyz: torch.Tensor, ranges):
        """Decoder inputs of a whole batch from an explicit feature matrix (rows of sample i = ``ranges[i]``): what
        forward_backbone returns for it -- the training iteration runs its no-grad click rounds on the training-mode
        backbone's features this way (engine.py:82-116 of the reference)."""
        lib = L.load()
        self.refresh_decoder_if_stale(check_versions=True)
        feats = feats128.to(self.device, torch.float32).contiguous()
        raw = raw_xyz.to(self.device, torch.float32).contiguous()
        n = feats.shape[0]
        C4 = torch.zeros((n, 4), dtype=torch.int32, device=self.device)
        st = _SceneState()
        st.engine_id = id(self)
        st.ranges = [(int(s), int(e)) for s, e in ranges]
        with torch.no_grad():
            for b, (s, e) in enumerate(st.ranges):
                C4[s:e, 0] = b
            st.posenc, st.minmax = self._posenc_batch(raw, st.ranges)
        pcd = SparseTensor(features=feats, coordinates=C4)
        pcd._a3d = st
        coordinates = SparseTensor(features=raw, coordinates=C4)
        return pcd, None, coordinates, [[[None] * len(st.ranges)] for _ in range(4)] + [[list(st.posenc)]]

    def _posenc_batch(self, raw, ranges):
        """Fourier position encodings of every sample of a batch (agile3d.py:141-161 loops over the samples; each has its
        own min / max) in three launches: (list of [n_b, 128] views of one matrix, list of [6] min/max views)."""
        lib = L.load()
        ns = len(ranges)
        if ns > 64 or any(ranges[i][1] != ranges[i + 1][0] for i in range(ns - 1)) or any(e <= s for s, e in ranges):
            pes, mms = [], []          # unusual layouts: sample by sample
            tmp = torch.empty(256 * 6 * 4, dtype=torch.uint8, device=self.device)
            for (s, e) in ranges:
                pe = torch.empty((e - s, 128), dtype=torch.float32, device=self.device)
                mm = torch.empty(6, dtype=torch.float32, device=self.device)
                L.check(lib.a3d_posenc_fourier(_ptr(raw[s:e]), e - s, self.decoder.gauss_B_ptr, _ptr(mm), _ptr(pe), _ptr(tmp),
                                               tmp.numel(), _stream()), "a3d_posenc_fourier")
                pes.append(pe)
                mms.append(mm)
            return pes, mms
        s0, e1 = ranges[0][0], ranges[-1][1]
        starts = (C.c_int64 * (ns + 1))(*([r[0] - s0 for r in ranges] + [e1 - s0]))
        pe_all = torch.empty((e1 - s0, 128), dtype=torch.float32, device=self.device)
        mm_all = torch.empty((ns, 6), dtype=torch.float32, device=self.device)
        tmp = torch.empty(lib.a3d_posenc_batch_workspace_bytes(ns), dtype=torch.uint8, device=self.device)
        L.check(lib.a3d_posenc_fourier_batch(_ptr(raw[s0:e1]), starts, ns, self.decoder.gauss_B_ptr, _ptr(mm_all), _ptr(pe_all),
                                             _ptr(tmp), tmp.numel(), _stream()), "a3d_posenc_fourier_batch")
        return [pe_all[s - s0:e - s0] for (s, e) in ranges], [mm_all[b] for b in range(ns)]

    # ---------------------------------------------------------------- forward_mask
    def forward_mask(self, pcd_features, aux, coordinates, pos_encodings_pcd, click_idx=None, click_time_idx=None):
        lib = L.load()
        st = getattr(pcd_features, "_a3d", None)
        if st is None or st.engine_id != id(self):
            raise RuntimeError("forward_mask needs the objects returned by this model's forward_backbone")
        if click_idx is None or click_time_idx is None:
            raise ValueError("click_idx and click_time_idx are required")
        self.refresh_decoder_if_stale(check_versions=True)
        W = self.decoder.W
        n_layers = self.decoder.n_layers
        preds = [[] for _ in range(n_layers)]
        # The interactive loop calls forward_mask ~100 times on one backbone output (eval_multi_obj.py:112-160): from the
        # second call on the scene's first-layer keys / values / scene-to-click queries are kept (123 MB per 80 k voxels; A3D_KV_CACHE_MB caps the
        # total, 0 switches the cache off).  A single call per scene -- the throughput benchmark -- allocates nothing.
        st.mask_calls += 1
        kv_state = 0
        kv_fill = None
        # what the cached keys / values were made from: the decoder pack AND the feature rows (a tensor swapped or
        # written in place between two calls must not be served stale keys)
        F_ = pcd_features.F
        kv_key = (self._dec_epoch, F_.data_ptr(), int(F_._version), tuple(st.ranges))
        if st.mask_calls >= 2 and _kv_cache_mb() > 0:
            if st.kv0 is not None and st.kv0_version == kv_key:
                kv_state = 2
            else:
                st.kv0, st.kv0_version = None, None
                total = sum(e - s for s, e in st.ranges) * 3 * 128 * 4      # keys, values, scene-to-click queries of the first layer
                if total <= _kv_cache_mb() * (1 << 20):
                    kv_fill = [torch.empty((3, e - s, 128), dtype=torch.float32, device=self.device) for s, e in st.ranges]
                    kv_state = 1
        kv_bufs = st.kv0 if kv_state == 2 else kv_fill
        with torch.no_grad():
            # the reference loops over the batch samples (agile3d.py:192); here every sample is described once and
            # the whole batch goes through a3d_decoder_forward_batch (one launch of each wide kernel per layer)
            samples = (L.DecoderSample * len(st.ranges))()
            keep = []                                   # host arrays / tensors the library reads during the call
            for b, (s, e) in enumerate(st.ranges):
                nb = e - s
                ci, ct = click_idx[b], click_time_idx[b]
                K = len(ci) - 1
                rows, objs, times = [], [], []
                for o in list(range(1, K + 1)) + [0]:
                    r, t = ci[str(o)], ct[str(o)]
                    if len(r) != len(t):
                        raise ValueError("click_idx / click_time_idx length mismatch")
                    if o > 0 and len(r) == 0:
                        raise ValueError(f"object {o} has no click (the reference fails on an empty max, "
                                         "agile3d.py:353)")
                    rows.extend(map(int, r))
                    times.extend(map(int, t))
                    objs.extend([o] * len(r))
                nc = len(rows)
                nq = nc + W.n_bg_queries
                wsb = lib.a3d_decoder_workspace_bytes(nb, nq)
                if wsb == 0:
                    raise L.A3DError(f"too many queries ({nq} > {L.A3D_MAX_QUERIES})")
                ws = torch.empty(wsb, dtype=torch.uint8, device=self.device)
                logits = torch.empty((n_layers, nb, K + 1), dtype=torch.float32, device=self.device)
                arr = lambda v: (C.c_int32 * max(1, len(v)))(*v)
                feats = pcd_features.F.detach()[s:e]
                a_rows, a_objs, a_times = arr(rows), arr(objs), arr(times)
                keep += [ws, feats, a_rows, a_objs, a_times]
                sp = samples[b]
                sp.feats128_dev, sp.posenc_dev, sp.n = _ptr(feats), _ptr(st.posenc[b]), nb
                sp.click_row = C.cast(a_rows, C.POINTER(C.c_int32))
                sp.click_obj = C.cast(a_objs, C.POINTER(C.c_int32))
                sp.click_time = C.cast(a_times, C.POINTER(C.c_int32))
                sp.n_clicks, sp.n_objects = nc, K
                sp.logits_dev, sp.workspace_dev, sp.workspace_bytes = _ptr(logits), _ptr(ws), wsb
                sp.kv0_dev, sp.kv0_state = (_ptr(kv_bufs[b]), kv_state) if kv_state else (None, 0)
                sp.kv0_blocks = int(kv_bufs[b].shape[0]) if kv_state else 0
                for l in range(n_layers):
                    preds[l].append(logits[l])
            try:
                L.check(lib.a3d_decoder_forward_batch(C.byref(W), samples, len(st.ranges), _stream()),
                        "a3d_decoder_forward_batch")
            except Exception:
                st.kv0, st.kv0_version = None, None     # a failed call leaves no cache behind (filled or being read)
                raise
            if kv_state == 1:                           # stamped valid only once the fill has been issued without an error
                st.kv0, st.kv0_version = kv_fill, kv_key
        out = {"pred_masks": preds[-1], "backbone_features": pcd_features}
        if self.model.aux:
            out["aux_outputs"] = [{"pred_masks": p} for p in preds[:-1]]
        return out


class _AuxFeature:
    """One of the 5 backbone feature maps (res16unet.py:250-290), rows in the library's internal
    order with matching coordinates.  The reference's forward_mask never reads ``aux``."""

    def __init__(self, program, st, i):
        self._p, self._st, self._i = program, st, i

    @property
    def F(self):
        return self._p.feature_map(self._st.scene, self._st.ws, self._i)

    @property
    def C(self):
        level = 4 - self._i
        t = torch.from_numpy(self._st.scene.table(level, L.TAB_XYZB).reshape(-1, 4).astype(np.int32))
        scale = 1 << level
        c = torch.stack([t[:, 3], t[:, 0] * scale, t[:, 1] * scale, t[:, 2] * scale], 1)
        return c.to(self._st.ws.device)

    @property
    def device(self):
        return self._st.ws.device


class _AuxList(list):
    def __init__(self, program, st):
        super().__init__(_AuxFeature(program, st, i) for i in range(5))
