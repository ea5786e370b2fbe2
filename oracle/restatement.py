"""ORACLE (test infrastructure, never shipped in the product path).

Explicit-arithmetic CPU restatement of the hot path in the *kernel's* decomposition:
encoder -> memory K/V projection -> KV-cached autoregressive decoder + head.  It uses no
``torch.nn`` layer, only matmul / softmax / mean / var, works in fp32 or fp64, and returns the
intermediates the HIP kernels expose, so a kernel bug can be localised.  It is pinned against
the golden vectors through ``tests/test_oracle_golden.py`` (its outputs must equal the
reference-form oracle, which is pinned to the real reference).

Why the KV cache is a legal restatement of reference mgcvae.py:432-446 (full recompute on a
growing target): the decoder self-attention is causal, so hidden state j of every layer depends
on target tokens <= j only; recomputing it at a later step yields the same value.

Weights are a flat ``{name: tensor}`` dict with state_dict names, e.g.
``enc.transformer_encoder.layers.0.self_attn.in_proj_weight`` where the prefixes are
enc / dec / head / fuse_map / fuse_lane / cnn / lane / vit.
"""
from __future__ import annotations

import math

import torch

STD = 80.0
LN_EPS = 1e-5
NEG_SLOPE = 0.2


def flatten_modules(model_dict, node_type="VEHICLE", dtype=torch.float64):
    """registrar ModuleDict -> flat dict with short prefixes."""
    alias = {
        "node_history_encoder": "enc", "decoder/transformer_decoder": "dec", "decoder/MLP": "head",
        "fusion/hist_map": "fuse_map", "fusion/lane_hist": "fuse_lane", "map_encoder": "map",
        "lane_encoder": "lane", "Lane/MLP_Softmax": "lane_cls",
    }
    w = {}
    for key, mod in model_dict.items():
        short = alias[key.split("/", 1)[1]]
        for pn, t in mod.state_dict().items():
            w[f"{short}.{pn}"] = t.detach().to(dtype)
    return w


def linear(x, w, b=None):
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def layer_norm(x, g, b, eps=LN_EPS):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def mha_scores(q, k, v, nhead, bias=None):
    """q (B,Lq,d), k/v (B,Lk,d); bias broadcastable to (B,H,Lq,Lk) with 0 / -inf."""
    B, Lq, d = q.shape
    Lk = k.shape[1]
    dh = d // nhead
    qh = q.view(B, Lq, nhead, dh).transpose(1, 2) * (1.0 / math.sqrt(dh))
    kh = k.view(B, Lk, nhead, dh).transpose(1, 2)
    vh = v.view(B, Lk, nhead, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if bias is not None:
        s = s + bias
    p = torch.softmax(s, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, d)


def encoder_forward(w, x_st, nhead, prefix="enc"):
    """Transformer_model.py:45-59 -> memory (B,T,d).  Causal + key-padding masked, post-norm, ReLU."""
    B, T, _ = x_st.shape
    pad = torch.isnan(x_st)[:, :, 1]
    h = linear(torch.nan_to_num(x_st, nan=0.0), w[f"{prefix}.input_fc.weight"], w[f"{prefix}.input_fc.bias"])
    h = h + w[f"{prefix}.pos_encoder.pe"][:T]
    d = h.shape[-1]
    neg = torch.full((T, T), float("-inf"), dtype=h.dtype)
    bias = torch.triu(neg, diagonal=1)[None, None] + torch.where(pad, float("-inf"), 0.0).to(h.dtype)[:, None, None, :]
    L = 0
    while f"{prefix}.transformer_encoder.layers.{L}.linear1.weight" in w:
        p = f"{prefix}.transformer_encoder.layers.{L}."
        qkv = linear(h, w[p + "self_attn.in_proj_weight"], w[p + "self_attn.in_proj_bias"])
        a = mha_scores(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], nhead, bias)
        h = layer_norm(h + linear(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"]),
                       w[p + "norm1.weight"], w[p + "norm1.bias"])
        f = linear(torch.relu(linear(h, w[p + "linear1.weight"], w[p + "linear1.bias"])),
                   w[p + "linear2.weight"], w[p + "linear2.bias"])
        h = layer_norm(h + f, w[p + "norm2.weight"], w[p + "norm2.bias"])
        L += 1
    return h, pad


def memory_kv(w, memory, prefix="dec"):
    """Cross-attention K/V of every decoder layer, computed once (the reference redoes it per step)."""
    d = memory.shape[-1]
    out = []
    L = 0
    while f"{prefix}.transformer_decoder.layers.{L}.linear1.weight" in w:
        p = f"{prefix}.transformer_decoder.layers.{L}.multihead_attn."
        kv = linear(memory, w[p + "in_proj_weight"][d:], w[p + "in_proj_bias"][d:])
        out.append((kv[..., :d], kv[..., d:]))
        L += 1
    return out


def mlp_stack(w, x, prefix):
    """MLP.py:6-30 regression form: (Linear, LayerNorm, ReLU)* + Linear."""
    i = 0
    while f"{prefix}.layer_seq.lmlp_{i}.0.weight" in w:
        p = f"{prefix}.layer_seq.lmlp_{i}."
        x = torch.relu(layer_norm(linear(x, w[p + "0.weight"], w[p + "0.bias"]), w[p + "1.weight"], w[p + "1.bias"]))
        i += 1
    return linear(x, w[f"{prefix}.layer_seq.out_mlp.weight"], w[f"{prefix}.layer_seq.out_mlp.bias"])


def decoder_ar(w, memory, pad, start_pos, nhead, ph, map_feat=None, feedback=True, tgt=None, prefix="dec"):
    """KV-cached form of mgcvae.py:425-447.  Returns (positions (B,ph,2) standardised, h (B,ph,nout)).

    feedback=False runs the same kernel teacher-forced on ``tgt`` (B,ph,2) and returns the per-token
    decoder outputs (Trajectory_Decoder.forward, Transformer_model.py:168-199).
    """
    B, T, d = memory.shape
    mkv = memory_kv(w, memory, prefix)
    cross_bias = torch.where(pad, float("-inf"), 0.0).to(memory.dtype)[:, None, None, :]
    nl = len(mkv)
    kc = [torch.zeros(B, ph, d, dtype=memory.dtype) for _ in range(nl)]
    vc = [torch.zeros(B, ph, d, dtype=memory.dtype) for _ in range(nl)]
    pos = start_pos.clone() if feedback else None
    outs, hs = [], []
    for s in range(ph):
        tok = pos if feedback else tgt[:, s]
        h = linear(tok, w[f"{prefix}.input_fc.weight"], w[f"{prefix}.input_fc.bias"]) + w[f"{prefix}.pos_encoder.pe"][s]
        h = h[:, None, :]
        for L in range(nl):
            p = f"{prefix}.transformer_decoder.layers.{L}."
            qkv = linear(h, w[p + "self_attn.in_proj_weight"], w[p + "self_attn.in_proj_bias"])
            kc[L][:, s] = qkv[:, 0, d:2 * d]
            vc[L][:, s] = qkv[:, 0, 2 * d:]
            a = mha_scores(qkv[..., :d], kc[L][:, :s + 1], vc[L][:, :s + 1], nhead)
            h = layer_norm(h + linear(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"]),
                           w[p + "norm1.weight"], w[p + "norm1.bias"])
            q = linear(h, w[p + "multihead_attn.in_proj_weight"][:d], w[p + "multihead_attn.in_proj_bias"][:d])
            a = mha_scores(q, mkv[L][0], mkv[L][1], nhead, cross_bias)
            h = layer_norm(h + linear(a, w[p + "multihead_attn.out_proj.weight"], w[p + "multihead_attn.out_proj.bias"]),
                           w[p + "norm2.weight"], w[p + "norm2.bias"])
            f = linear(torch.relu(linear(h, w[p + "linear1.weight"], w[p + "linear1.bias"])),
                       w[p + "linear2.weight"], w[p + "linear2.bias"])
            h = layer_norm(h + f, w[p + "norm3.weight"], w[p + "norm3.bias"])
        ho = linear(h[:, 0], w[f"{prefix}.output_fc.weight"], w[f"{prefix}.output_fc.bias"])
        hs.append(ho)
        if feedback:
            z = ho
            if map_feat is not None:
                z = mlp_stack(w, torch.cat([z, map_feat], dim=-1), "fuse_map")
            pos = pos + mlp_stack(w, z, "head")
            outs.append(pos)
    h_all = torch.stack(hs, dim=1)
    return (torch.stack(outs, dim=1) if feedback else None), h_all


def leaky(x):
    return torch.where(x > 0, x, NEG_SLOPE * x)


def cnn_map_forward(w, maps, strides=(2, 2, 1, 1), prefix="map"):
    """map_encoder.py:31-36 on ``map*2-1`` (mgcvae.py:290); conv as unfold + matmul."""
    x = maps * 2.0 - 1.0
    i = 0
    while f"{prefix}.convs.{i}.weight" in w:
        wt, b = w[f"{prefix}.convs.{i}.weight"], w[f"{prefix}.convs.{i}.bias"]
        co, ci, k, _ = wt.shape
        stride = strides[i]
        cols = torch.nn.functional.unfold(x, k, stride=stride)              # (B, ci*k*k, L)
        ho = (x.shape[-2] - k) // stride + 1
        y = (wt.reshape(co, -1) @ cols + b[None, :, None]).view(x.shape[0], co, ho, -1)
        x = leaky(y)
        i += 1
    return linear(x.flatten(1), w[f"{prefix}.fc.weight"], w[f"{prefix}.fc.bias"])


def lane_forward(w, lanes, stride=2, prefix="lane"):
    """Lane_encoder.py:29-33 on the raw ``.view(B*3, 2, 10)`` of (B,3,10,2) (mgcvae.py:301-302)."""
    B, ln, sn, emb = lanes.shape
    x = lanes.reshape(B * ln, emb, sn)            # same memory order reinterpretation as .view
    wt, b = w[f"{prefix}.convs.0.weight"], w[f"{prefix}.convs.0.bias"]     # (co, emb, k)
    co, _, k = wt.shape
    width = (sn - k) // stride + 1
    cols = torch.stack([x[:, :, j * stride:j * stride + k].reshape(B * ln, -1) for j in range(width)], dim=-1)
    y = leaky(torch.einsum("oc,ncw->now", wt.reshape(co, -1), cols) + b[None, :, None])
    return linear(y.flatten(1), w[f"{prefix}.fc.weight"], w[f"{prefix}.fc.bias"]).view(B, ln, -1)


def predict(w, hp, x, x_st, lanes=None, maps=None, ph=12):
    """mgcvae.py:720-773 in kernel decomposition; returns metres."""
    nhead = hp["transformer"]["nhead"]
    memory, pad = encoder_forward(w, x_st, nhead)
    start = x_st[:, -1, 0:2]
    node = x[:, -1, 0:2]
    if hp["lane_cnn_encoding"]:
        enc_lane = lane_forward(w, lanes, hp["lane_encoder"]["VEHICLE"]["strides"])
        T = memory.shape[1]
        feat0 = enc_lane[:, 0:1, :].expand(-1, T, -1)         # lane-0 reuse quirk (mgcvae.py:404)
        mem = mlp_stack(w, torch.cat([memory, feat0], dim=-1), "fuse_lane")
        pos, _ = decoder_ar(w, mem, pad, start, nhead, ph)
        out = pos * STD + node[:, None, :]
        return torch.stack([out, out, out], dim=1)
    feat = None
    if hp["map_cnn_encoding"]:
        feat = cnn_map_forward(w, maps, hp["map_encoder"]["cnn_param"]["strides"])
    pos, _ = decoder_ar(w, memory, pad, start, nhead, ph, map_feat=feat)
    return pos * STD + node[:, None, :]
