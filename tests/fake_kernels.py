"""TEST INFRASTRUCTURE ONLY: torch/CPU stand-ins for the eight training kernels of
``multimodal_transformer_amd/_train_kernels.py``, with the same argument conventions.

They exist so that the *host orchestration* in ``training.py`` (which buffer feeds which kernel, the reverse-time
order, the deferred weight-gradient GEMMs) can be checked against the reference's autograd goldens in the CPU
test tier.  The product never imports this file; on a GPU box the real HIP kernels are what run
(tests/test_gpu_training.py checks those against the same goldens).
"""
import numpy as np
import torch


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on uint32 arrays:
    counter (..., 4), key (2,) -> (..., 4).  The restatement the mask kernel (csrc/train.hip) is checked against."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, W0, W1, lo32 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), \
        np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & lo32, p1 >> np.uint64(32), p1 & lo32
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + W0) & lo32, (k1 + W1) & lo32
    return np.stack(c, axis=-1).astype(np.uint32)


def dropout_keep_bytes(n, p, seed, step, stream_id):
    """The keep bytes mmt_dropout_mask writes (include/mmt_amd.h): element i <- word i%4 of Philox(counter = (i/4 low, i/4 high
    ^ stream_id << 8, step low, step high), key = seed) >= p * 2^32."""
    n4 = (n + 3) // 4
    i = np.arange(n4, dtype=np.uint64)
    ctr = np.stack([i & np.uint64(0xFFFFFFFF), (i >> np.uint64(32)) ^ np.uint64((stream_id << 8) & 0xFFFFFFFF),
                    np.full(n4, step & 0xFFFFFFFF, dtype=np.uint64), np.full(n4, (step >> 32) & 0xFFFFFFFF, dtype=np.uint64)], axis=-1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    thr = np.uint32(min(4294967295.0, float(np.float32(p)) * 4294967296.0))
    return (r >= thr).astype(np.uint8).reshape(-1)[:n]


def dropout_mask(keep, p, rng_state, stream_id):
    b = dropout_keep_bytes(keep.numel(), p, int(rng_state[0]), int(rng_state[1]), stream_id)
    keep.copy_(torch.from_numpy(b).view(keep.shape))


def rng_advance(rng_state):
    rng_state[1] += 1


def dropout_apply(x, keep, scale, y):
    y.copy_(torch.where(keep.view(x.shape) != 0, x * scale, torch.zeros_like(x)))


def _keepf(drop):
    keep, scale = drop[0], drop[1]
    return (keep != 0).double() * scale


def gemm(A, B, Cm, bias=None, addend=None, mask=None, relu=False, accumulate=False, ln_bwd=None, splitk=1, part_stride=0,
         asum=None, drop=None, drop_after_add=False, mask_scale=0.0, ln_drop=None):
    # (split-K partial buffers: everything lands in slice 0, the other slices keep their zeros)
    if asum is not None:
        asum.copy_(A.double().sum(1).to(asum.dtype))
    if ln_bwd is not None:
        xhat, rstd, gamma, dx = ln_bwd
        layernorm_bwd(A, xhat, rstd, gamma, dx, drop=ln_drop)
        A = dx if ln_drop is None else ln_drop[2]
    v = A.double() @ B.double()
    if bias is not None:
        v = v + bias.double()
    if drop is not None:
        if drop_after_add:
            if addend is not None:
                v = v + addend.double()
            v = v * _keepf(drop)
        else:
            if relu:
                v = torch.relu(v)
            v = v * _keepf(drop)
            if addend is not None:
                v = v + addend.double()
    else:
        if addend is not None:
            v = v + addend.double()
        if relu:
            v = torch.relu(v)
    if mask is not None:
        v = torch.where(mask > 0, v * (mask_scale if mask_scale else 1.0), torch.zeros_like(v))
    v = v.to(Cm.dtype)
    if accumulate:
        Cm.add_(v)
    else:
        Cm.copy_(v)


def colsum(a, out, b=None, mask=None, accumulate=False):
    v = a.double()
    if b is not None:
        v = v * b.double()
    if mask is not None:
        v = torch.where(mask > 0, v, torch.zeros_like(v))
    v = v.sum(0).to(out.dtype)
    if accumulate:
        out.add_(v)
    else:
        out.copy_(v)


def _as_nchw(inp, B, C, H, W, nhwc):
    return inp[:, :C].reshape(B, H, W, C).permute(0, 3, 1, 2) if nhwc else inp


def im2col(inp, B, C, H, W, kh, kw, stride, cols, nhwc, alpha=1.0, beta=0.0, slope=1.0):
    v = _as_nchw(inp, B, C, H, W, nhwc).double() * alpha + beta
    v = torch.where(v > 0, v, slope * v)
    u = torch.nn.functional.unfold(v, (kh, kw), stride=stride)            # (B, C*kh*kw, OH*OW), column order (c, ky, kx)
    cols.copy_(u.permute(0, 2, 1).reshape(cols.shape).to(cols.dtype))


def col2im(dcols, B, C, H, W, kh, kw, stride, din, z=None, slope=1.0):
    OH, OW = (H - kh) // stride + 1, (W - kw) // stride + 1
    u = dcols.double().reshape(B, OH * OW, C * kh * kw).permute(0, 2, 1)
    g = torch.nn.functional.fold(u, (H, W), (kh, kw), stride=stride)      # (B, C, H, W)
    g = g.permute(0, 2, 3, 1).reshape(B * H * W, C)
    if z is not None:
        g = g * torch.where(z > 0, torch.ones_like(g), torch.full_like(g, slope))
    din.copy_(g.to(din.dtype))


def ln_param_grads(dy, xhat, gw, gb, splitk=1, part_stride=0):
    gw.copy_((dy.double() * xhat.double()).sum(0).to(gw.dtype))
    gb.copy_(dy.double().sum(0).to(gb.dtype))


def gemm_group(calls, cache=None):
    for kw in calls:
        gemm(**kw)


def ln_param_grads_group(calls, cache=None):
    for kw in calls:
        ln_param_grads(**kw)


def layernorm_fwd(x, res, g, b, y, xhat, rstd, relu=False):
    v = x.double() + (res.double() if res is not None else 0.0)
    mu = v.mean(-1, keepdim=True)
    var = ((v - mu) ** 2).mean(-1, keepdim=True)
    r = 1.0 / torch.sqrt(var + 1e-5)
    h = (v - mu) * r
    o = h * g.double() + b.double()
    if relu:
        o = torch.relu(o)
    y.copy_(o.to(y.dtype))
    xhat.copy_(h.to(xhat.dtype))
    rstd.copy_(r[:, 0].to(rstd.dtype))


def layernorm_bwd(dy, xhat, rstd, g, dx, y_relu=None, dy_masked=None, accumulate=False, drop=None):
    d = dy.double()
    if y_relu is not None:
        d = torch.where(y_relu > 0, d, torch.zeros_like(d))
    if dy_masked is not None:
        dy_masked.copy_(d.to(dy_masked.dtype))
    dg = d * g.double()
    xh = xhat.double()
    m1 = dg.mean(-1, keepdim=True)
    m2 = (dg * xh).mean(-1, keepdim=True)
    v = (rstd.double()[:, None] * (dg - m1 - xh * m2)).to(dx.dtype)
    if accumulate:
        dx.add_(v)
    else:
        dx.copy_(v)
    if drop is not None:
        drop[2].copy_((v.double() * _keepf(drop)).to(dx.dtype))


def _heads(t, H):
    B, L, d = t.shape
    return t.double().reshape(B, L, H, d // H).permute(0, 2, 1, 3)          # (B,H,L,dh)


def _strided(buf, B, nhead, Lq, Lk, probs_strides):
    hs, rs = probs_strides if probs_strides is not None else (Lq * Lk, Lk)
    return torch.as_strided(buf, (B, nhead, Lq, Lk), (nhead * hs, hs, rs, 1))


def attention_fwd(q, k, v, o, probs, nhead, scale, pad=None, causal=False, causal_offset=0, probs_strides=None, pdrop=None):
    B, Lq, d = q.shape
    Lk = k.shape[1]
    s = _heads(q, nhead) @ _heads(k, nhead).transpose(-1, -2) * scale       # (B,H,Lq,Lk)
    keep = torch.ones(B, 1, Lq, Lk, dtype=torch.bool)
    if causal:
        i = torch.arange(Lq)[:, None]
        j = torch.arange(Lk)[None, :]
        keep = keep & (j < i + 1 + causal_offset)[None, None]
    if pad is not None:
        keep = keep & ~(pad.bool()[:, None, None, :])
    s = s.masked_fill(~keep, float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)                                       # fully masked row -> zeros (kernel convention)
    _strided(probs, B, nhead, Lq, Lk, probs_strides).copy_(p.to(probs.dtype))
    if pdrop is not None:
        p = p * (_strided(pdrop[0], B, nhead, Lq, Lk, probs_strides) != 0).double() * pdrop[1]
    out = (p @ _heads(v, nhead)).permute(0, 2, 1, 3).reshape(B, Lq, d)
    o.copy_(out.to(o.dtype))


def attention_bwd(q, k, v, probs, d_o, dq, dk, dv, nhead, scale, probs_strides=None, pdrop=None):
    B, Lq, d = q.shape
    Lk = k.shape[1]
    p = _strided(probs, B, nhead, Lq, Lk, probs_strides).double()
    keep = torch.ones_like(p)
    if pdrop is not None:
        keep = (_strided(pdrop[0], B, nhead, Lq, Lk, probs_strides) != 0).double() * pdrop[1]
    g = _heads(d_o, nhead)
    dp = (g @ _heads(v, nhead).transpose(-1, -2)) * keep
    ds = p * (dp - (p * dp).sum(-1, keepdim=True)) * scale
    merge = lambda t, L: t.permute(0, 2, 1, 3).reshape(B, L, d)
    dq.copy_(merge(ds @ _heads(k, nhead), Lq).to(dq.dtype))
    dk.add_(merge(ds.transpose(-1, -2) @ _heads(q, nhead), Lk).to(dk.dtype))
    dv.add_(merge((p * keep).transpose(-1, -2) @ g, Lk).to(dv.dtype))


def gemm_attention_bwd_ok(A, nhead, Lk, dh):
    return True


def gemm_attention_bwd(A, B, q, k, v, probs, dq, dk, dv, nhead, scale, ln_bwd=None, probs_strides=None, ln_drop=None, pdrop=None):
    d_o = torch.zeros(A.shape[0], B.shape[1], dtype=A.dtype)
    gemm(A, B, d_o, ln_bwd=ln_bwd, ln_drop=ln_drop)
    attention_bwd(q, k, v, probs, d_o.view(A.shape[0], 1, -1), dq, dk, dv, nhead, scale, probs_strides=probs_strides, pdrop=pdrop)


def l2_loss_bwd(pos, y, dpos, loss_sum=None):
    ph, B, _ = pos.shape
    diff = pos.double() - y.double().permute(1, 0, 2)
    n = diff.norm(dim=-1, keepdim=True)
    w = torch.where(n > 0, 1.0 / (n * B * ph), torch.zeros_like(n))
    dpos.copy_((diff * w).to(dpos.dtype))
    if loss_sum is not None:
        loss_sum.add_((n.sum() / (B * ph)).to(loss_sum.dtype))


def history_prepare(x_st, x0, pad):
    B, T, n = x_st.shape
    x0.copy_(torch.nan_to_num(x_st, nan=0.0).reshape(B * T, n))
    pad.copy_(torch.isnan(x_st)[:, :, 1].to(torch.uint8))


def adamw_table(entries, device):
    return list(entries), len(entries)


def adamw_step(table, ntensors, state, beta1, beta2, eps, weight_decay, clip, grad_scale, guard=None):
    if guard is not None and not bool(torch.isfinite(guard).all()):
        return
    step, lr0 = float(state[0]), float(state[1])
    for p, g, m, v, s in table:
        lr = lr0 * s
        gg = g * grad_scale
        if clip > 0:
            gg = gg.clamp(-clip, clip)
        p.mul_(1.0 - lr * weight_decay)
        m.mul_(beta1).add_(gg, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1.0 - beta2)
        denom = v.sqrt() / (1.0 - beta2 ** step) ** 0.5 + eps
        p.addcdiv_(m, denom, value=-lr / (1.0 - beta1 ** step))


def lr_step(state, gamma, guard=None, skipped=None):
    if guard is not None and not bool(torch.isfinite(guard).all()):
        if skipped is not None:
            skipped[0] += 1.0
        return
    state[0] += 1.0
    state[1] *= gamma


def softmax_axis(x, y, axis):
    y.copy_(torch.softmax(x.double(), dim=axis).to(y.dtype))


def cumsum_axis(x, y, axis):
    y.copy_(torch.cumsum(x.double(), dim=axis).to(y.dtype))


def add_inplace(y, x):
    y.add_(x)


def ar_decode(eng):
    """Stand-in for TrainEngine._ar_decode (the fused HIP decoder): the same autoregressive recurrence, obtained by
    running the engine's own all-steps pass once per step with the teacher-forced wiring (causal attention makes step s
    independent of the tokens after it) and feeding position s+1 = position s + head output back as the next token."""
    S, ph = eng.s, eng.ph
    for s in range(ph):
        eng._decoder_steps(0, ph)
        eng.head_tape.forward_out(eng._head_z, S.delta.view(ph * eng.B, -1))
        S.pos[s + 1].copy_(S.pos[s] + S.delta[s])


ALL = ("gemm_group", "ln_param_grads_group", "dropout_mask", "rng_advance", "dropout_apply", "im2col", "col2im", "gemm_attention_bwd_ok", "gemm_attention_bwd", "ln_param_grads", "softmax_axis", "cumsum_axis", "add_inplace", "gemm", "colsum", "layernorm_fwd", "layernorm_bwd", "attention_fwd", "attention_bwd", "l2_loss_bwd",
       "history_prepare", "adamw_table", "adamw_step", "lr_step")


def install(monkeypatch=None):
    """Patch the kernel wrappers (and the engine's fused-decoder call) with the stand-ins; without a monkeypatch
    fixture (spawned worker processes) the patch is permanent for the process."""
    import sys
    from multimodal_transformer_amd import _train_kernels as K
    from multimodal_transformer_amd.training import TrainEngine
    me = sys.modules[__name__]
    put = monkeypatch.setattr if monkeypatch is not None else setattr
    for n in ALL:
        put(K, n, torch.no_grad()(getattr(me, n)))   # the real kernels never build autograd graphs
    put(TrainEngine, "_ar_decode", torch.no_grad()(ar_decode))
