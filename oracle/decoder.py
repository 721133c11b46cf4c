"""CPU restatement of the reference's click-query decoder (TEST INFRASTRUCTURE).

Follows ``models/agile3d.py:183-384`` (forward_mask, mask_module),
``models/modules/attention_block.py:28-38,86-98,151-155`` (post-norm layers over
nn.MultiheadAttention), ``models/position_embedding.py:13-41,123-152,210-225``.
PINNED against the reference's own code: see tests/golden/make_goldens.py and
tests/test_oracle_decoder.py (max abs diff <= 1e-5 on every golden case).

All functions take the model ``state_dict`` (reference key layout) and plain tensors.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5


# ----------------------------------------------------------------------------- encodings
def fourier_pos_enc(xyz, gauss_B, mins, maxs):
    """PositionEmbeddingCoordsSine.get_fourier_embeddings with normalize=True
    (position_embedding.py:123-152, shift_scale_points :13-41).  xyz [n,3] -> [n,128]."""
    xyz = xyz.float()
    src_diff = (maxs - mins).reshape(1, 3)
    u = ((xyz - mins.reshape(1, 3)) * 1.0) / src_diff + 0.0
    u = u * (2 * np.pi)
    proj = u @ gauss_B
    return torch.cat([proj.sin(), proj.cos()], dim=1)


def time_table(d_model: int = 128, length: int = 200):
    """PositionalEncoding1D, position_embedding.py:210-225."""
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


# ----------------------------------------------------------------------------- layers
# the activation of the FFN and of the mask MLP: a module attribute so that the gradient comparison of the training path
# can substitute the 0/1 masks of the implementation under test (see oracle/backbone.py: RELU)
RELU = torch.relu


def layer_norm(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"], LN_EPS)


def mha(sd, prefix, query, key, value, attn_mask=None, nhead: int = 8):
    """nn.MultiheadAttention forward on unbatched [L,E] inputs (dropout 0).  attn_mask: bool
    [Lq,Lk], True = blocked."""
    E = query.shape[-1]
    W, b = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    q = query @ W[:E].T + b[:E]
    k = key @ W[E:2 * E].T + b[E:2 * E]
    v = value @ W[2 * E:].T + b[2 * E:]
    dh = E // nhead
    Lq, Lk = q.shape[0], k.shape[0]
    q = q.reshape(Lq, nhead, dh).transpose(0, 1) * (1.0 / math.sqrt(dh))
    k = k.reshape(Lk, nhead, dh).transpose(0, 1)
    v = v.reshape(Lk, nhead, dh).transpose(0, 1)
    s = q @ k.transpose(1, 2)  # [h, Lq, Lk]
    if attn_mask is not None:
        s = s.masked_fill(attn_mask.unsqueeze(0), float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(0, 1).reshape(Lq, E)
    return o @ sd[prefix + "out_proj.weight"].T + sd[prefix + "out_proj.bias"]


def cross_attention_layer(sd, prefix, tgt, memory, memory_mask, pos, query_pos):
    """CrossAttentionLayer.forward_post, attention_block.py:86-98."""
    tgt2 = mha(sd, prefix + "multihead_attn.", tgt + query_pos, memory + pos, memory, memory_mask)
    return layer_norm(tgt + tgt2, sd, prefix + "norm.")


def self_attention_layer(sd, prefix, tgt, query_pos):
    """SelfAttentionLayer.forward_post, attention_block.py:28-38."""
    qk = tgt + query_pos
    tgt2 = mha(sd, prefix + "self_attn.", qk, qk, tgt, None)
    return layer_norm(tgt + tgt2, sd, prefix + "norm.")


def ffn_layer(sd, prefix, tgt):
    """FFNLayer.forward_post, attention_block.py:151-155."""
    h = RELU(tgt @ sd[prefix + "linear1.weight"].T + sd[prefix + "linear1.bias"])
    tgt2 = h @ sd[prefix + "linear2.weight"].T + sd[prefix + "linear2.bias"]
    return layer_norm(tgt + tgt2, sd, prefix + "norm.")


def mask_module(sd, fg_q, bg_q, mask_features, fg_split):
    """Agile3d.mask_module, agile3d.py:342-384.  Returns (logits [N,1+K], attn_mask [Q,N] bool)."""
    def embed(q):
        q = layer_norm(q, sd, "decoder_norm.")
        h = RELU(q @ sd["mask_embed_head.0.weight"].T + sd["mask_embed_head.0.bias"])
        return h @ sd["mask_embed_head.2.weight"].T + sd["mask_embed_head.2.bias"]

    fg_prods = (mask_features @ embed(fg_q).T).split(fg_split, dim=1)
    fg_masks = torch.cat([p.max(dim=-1, keepdim=True)[0] for p in fg_prods], dim=-1)
    bg_masks = (mask_features @ embed(bg_q).T).max(dim=-1, keepdim=True)[0]
    out = torch.cat([bg_masks, fg_masks], dim=-1)
    labels = out.argmax(1)
    rows = []
    for obj in range(1, fg_masks.shape[-1] + 1):
        m = ~(labels == obj)
        if bool(m.all()):            # row that would be all True is reset to all False (:375)
            m = torch.zeros_like(m)
        rows.append(m.unsqueeze(0).repeat(fg_split[obj - 1], 1))
    mb = ~(labels == 0)
    if bool(mb.all()):
        mb = torch.zeros_like(mb)
    rows.append(mb.unsqueeze(0).repeat(bg_q.shape[0], 1))
    return out, torch.cat(rows, dim=0)


# ----------------------------------------------------------------------------- forward_mask
def forward_mask(sd, pcd_features, raw_xyz, pos_enc, click_idx, click_time_idx,
                 num_decoders: int = 3, return_masks: bool = False, grad: bool = False, force_masks=None):
    """Agile3d.forward_mask for ONE batch sample (agile3d.py:192-323).

    pcd_features [N,128], raw_xyz [N,3], pos_enc [N,128] (level-4 Fourier encoding),
    click_idx / click_time_idx: dict str -> list[int] ('0' = background).
    Returns list of ``num_decoders`` logits tensors [N,1+K] (last = 'pred_masks', earlier =
    'aux_outputs').  ``grad=True`` keeps the autograd graph (training-path tests); ``force_masks`` replaces the
    label-derived attention masks of the layers by given ones (they are not differentiated; a test passes the masks of
    the implementation under test so that both follow the same branch).
    """
    with torch.set_grad_enabled(grad):
        mins, maxs = raw_xyz.min(0)[0], raw_xyz.max(0)[0]
        K = len(click_idx) - 1
        fg_split = [len(click_idx[str(i)]) for i in range(1, K + 1)]
        fg_rows = [r for i in range(1, K + 1) for r in click_idx[str(i)]]
        fg_times = [t for i in range(1, K + 1) for t in click_time_idx[str(i)]]
        tt = time_table(pcd_features.shape[1], 200)
        B = sd["pos_enc.gauss_B"]
        fg_pos = fourier_pos_enc(raw_xyz[fg_rows], B, mins, maxs) + tt[fg_times]
        bg_pos = sd["bg_query_pos.weight"]
        bg_q = sd["bg_query_feat.weight"]
        bg_rows = list(click_idx["0"])
        if len(bg_rows):
            bpos = fourier_pos_enc(raw_xyz[bg_rows], B, mins, maxs) + tt[list(click_time_idx["0"])]
            bg_pos = torch.cat([bg_pos, bpos], 0)
            bg_q = torch.cat([bg_q, pcd_features[bg_rows]], 0)
        fg_q = pcd_features[fg_rows]
        n_fg, n_bg = fg_q.shape[0], bg_q.shape[0]
        qpos = torch.cat([fg_pos, bg_pos], 0)
        src = pcd_features
        attn_mask = None
        outs, masks = [], []
        for d in range(num_decoders):
            out = cross_attention_layer(sd, f"c2s_attention.{d}.0.", torch.cat([fg_q, bg_q], 0), src,
                                        attn_mask, pos_enc, qpos)
            out = self_attention_layer(sd, f"c2c_attention.{d}.0.", out, qpos)
            queries = ffn_layer(sd, f"ffn_attention.{d}.0.", out)
            src = cross_attention_layer(sd, f"s2c_attention.{d}.0.", src, queries, None, qpos, pos_enc)
            fg_q, bg_q = queries.split([n_fg, n_bg], 0)
            logits, attn_mask = mask_module(sd, fg_q, bg_q, src, fg_split)
            if force_masks is not None:
                attn_mask = force_masks[d]
            outs.append(logits)
            masks.append(attn_mask)
    return (outs, masks) if return_masks else outs
