"""Tensor-view front end of the training-step kernels (include/mmt_amd.h, "training ops").

Each function takes torch *views* (device memory, fp32), derives raw pointers and element strides from
them and launches one HIP kernel on torch's current stream.  Nothing here computes with torch: a CPU
tensor raises.  `training.py` is written purely against this module's eight entry points.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import MmtError


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dev(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise MmtError(f"{name}: tensor is on {t.device}; the training kernels need device memory (no CPU fallback)")
    if t.dtype != dtype:
        raise MmtError(f"{name}: dtype {t.dtype}, expected {dtype}")
    return t


def _rows(t, name):
    """(M,N) view with unit column stride -> (ptr, ld)."""
    _dev(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise MmtError(f"{name}: expected a 2-D view with unit column stride, got shape {tuple(t.shape)} "
                       f"strides {t.stride()}")
    return t.stride(0)


def _keep(t, name, shape):
    """A dropout keep mask: uint8 view of ``shape`` with unit column stride -> row stride in bytes."""
    _dev(t, name, torch.uint8)
    if tuple(t.shape) != tuple(shape) or (t.shape[-1] > 1 and t.stride(-1) != 1):
        raise MmtError(f"{name}: expected a uint8 view of shape {tuple(shape)} with unit column stride, got "
                       f"{tuple(t.shape)} {t.stride()}")
    return t.stride(0)


def _gemm_args(A, B, Cm, bias=None, addend=None, mask=None, relu=False, accumulate=False, ln_bwd=None, splitk=1,
               part_stride=0, asum=None, drop=None, drop_after_add=False, mask_scale=0.0, ln_drop=None):
    _dev(A, "A"), _dev(B, "B")
    M, K = A.shape
    K2, N = B.shape
    if K2 != K or (Cm is not None and tuple(Cm.shape) != (M, N)):
        raise MmtError(f"gemm: shapes A{tuple(A.shape)} B{tuple(B.shape)} C{tuple(Cm.shape) if Cm is not None else None} "
                       "do not agree")
    g = _lib.GemmArgs()
    g.A, g.sam, g.sak = A.data_ptr(), A.stride(0), A.stride(1)
    g.B, g.sbk, g.sbn = B.data_ptr(), B.stride(0), B.stride(1)
    if Cm is not None:
        g.C, g.ldc = Cm.data_ptr(), _rows(Cm, "C")
    g.M, g.N, g.K = M, N, K
    if bias is not None:
        _dev(bias, "bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise MmtError("gemm: bias must be a contiguous vector of N elements")
        g.bias = bias.data_ptr()
    if addend is not None:
        if tuple(addend.shape) != (M, N):
            raise MmtError(f"gemm: addend shape {tuple(addend.shape)} != {(M, N)}")
        g.addend, g.ld_add = addend.data_ptr(), _rows(addend, "addend")
    if mask is not None:
        if tuple(mask.shape) != (M, N):
            raise MmtError(f"gemm: mask shape {tuple(mask.shape)} != {(M, N)}")
        g.mask, g.ld_mask = mask.data_ptr(), _rows(mask, "mask")
    g.relu, g.accumulate = int(bool(relu)), int(bool(accumulate))
    if ln_bwd is not None:
        xhat, rstd, gamma, dx = ln_bwd
        _dev(xhat, "xhat"), _dev(rstd, "rstd"), _dev(gamma, "ln weight"), _dev(dx, "dx")
        if tuple(xhat.shape) != (M, K) or tuple(dx.shape) != (M, K) or _rows(xhat, "xhat") != _rows(dx, "dx"):
            raise MmtError("gemm: ln_bwd xhat / dx must be (M,K) with equal row strides")
        if rstd.numel() != M or not rstd.is_contiguous() or gamma.numel() != K or not gamma.is_contiguous():
            raise MmtError("gemm: ln_bwd rstd (M) / weight (K) must be contiguous vectors")
        g.ln_xhat, g.ln_rstd, g.ln_g, g.ln_dx, g.ld_ln = xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), \
            dx.data_ptr(), _rows(xhat, "xhat")
    if mask_scale:
        g.mask_scale = float(mask_scale)
    if drop is not None:            # (keep bytes (M,N), scale)
        keep, scale = drop
        g.drop, g.ld_drop, g.drop_scale = keep.data_ptr(), _keep(keep, "drop", (M, N)), float(scale)
        g.drop_after_add = int(bool(drop_after_add))
    if ln_drop is not None:         # (keep bytes (M,K), scale, masked copy (M,K) laid out like dx)
        if ln_bwd is None:
            raise MmtError("gemm: ln_drop goes with ln_bwd")
        keep, scale, dxm = ln_drop
        _dev(dxm, "dxm")
        if tuple(dxm.shape) != (M, K) or _rows(dxm, "dxm") != g.ld_ln:
            raise MmtError("gemm: ln_drop's masked copy must be laid out like dx")
        g.ln_drop, g.ld_ln_drop, g.ln_dxm, g.ln_drop_scale = keep.data_ptr(), _keep(keep, "ln_drop", (M, K)), dxm.data_ptr(), \
            float(scale)
    if splitk > 1 or asum is not None:
        if bias is not None or addend is not None or mask is not None or relu or accumulate or ln_bwd is not None \
                or drop is not None:
            raise MmtError("gemm: splitk / asum cannot be combined with an epilogue")
        g.splitk, g.part_stride = int(splitk), int(part_stride)
        if asum is not None:
            _dev(asum, "asum")
            if asum.numel() != M or not asum.is_contiguous():
                raise MmtError("gemm: asum must be a contiguous vector of M elements")
            g.asum = asum.data_ptr()
    return g


def gemm(A, B, Cm, bias=None, addend=None, mask=None, relu=False, accumulate=False, ln_bwd=None, splitk=1,
         part_stride=0, asum=None, drop=None, drop_after_add=False, mask_scale=0.0, ln_drop=None):
    """Cm (+)= epi(A' @ B): A (M,K) and B (K,N) are arbitrary-stride views, Cm (M,N) has unit column stride.
    epi(v) = v + bias[n] + addend[m,n] -> relu -> zero where mask[m,n] <= 0.
    ``ln_bwd=(xhat, rstd, g, dx)``: A is the gradient at a LayerNorm output and A' = its LayerNorm backward (the
    input gradient, also stored to ``dx``), fused into the operand load; without it A' = A.
    ``splitk`` > 1 (weight gradients): slice z of the K range writes its partial to ``Cm`` + z*``part_stride`` floats
    (same for ``asum``), the caller sums the slices; ``asum`` (M) receives the row sums of A (the bias gradient).
    ``drop=(keep, scale)`` (training-mode dropout, keep = uint8 (M,N)): epi(v) = keep*scale*[relu](v + bias) + addend, or
    keep*scale*(v + bias + addend) with ``drop_after_add``.  ``mask_scale``: outputs passing ``mask`` are multiplied by it.
    ``ln_drop=(keep, scale, dxm)`` with ``ln_bwd``: the multiplied operand is keep*scale*A', also stored to ``dxm``."""
    g = _gemm_args(A, B, Cm, bias, addend, mask, relu, accumulate, ln_bwd, splitk, part_stride, asum, drop, drop_after_add,
                   mask_scale, ln_drop)
    lib = _lib.load()
    _lib.check(lib.mmt_gemm(C.byref(g), _stream()), "mmt_gemm")


def gemm_attention_bwd_ok(A, nhead, Lk, dh):
    """Shapes mmt_gemm_attention_bwd takes: the per-step sweep at batch <= 256, head dim 32, <= 16 keys."""
    M, K = A.shape
    return M <= 256 and K % 16 == 0 and K <= 256 and dh == 32 and 1 <= Lk <= 16


def gemm_attention_bwd(A, B, q, k, v, probs, dq, dk, dv, nhead, scale, ln_bwd=None, probs_strides=None, ln_drop=None,
                       pdrop=None):
    """d_o = A' @ B (A' as in ``gemm``, no epilogue) is consumed in place by ``attention_bwd`` with one query per batch
    element: dq (=), dk (+=), dv (+=); d_o itself is not stored."""
    g = _gemm_args(A, B, None, ln_bwd=ln_bwd, ln_drop=ln_drop)
    a = _attn_args(q, k, v, probs, None, nhead, False, 0, scale, probs_strides=probs_strides, pdrop=pdrop)
    Bn, Lq, d = q.shape
    Lk = k.shape[1]
    if Lq != 1 or g.M != Bn or g.N != d:
        raise MmtError("gemm_attention_bwd: one query per batch element, GEMM output (B, d)")
    dq_bs, _ = _bld(dq, "dq", Bn, 1, d)
    dk_bs, dk_rs = _bld(dk, "dk", Bn, Lk, d)
    if _bld(dv, "dv", Bn, Lk, d) != (dk_bs, dk_rs):
        raise MmtError("gemm_attention_bwd: dk and dv must share strides")
    lib = _lib.load()
    _lib.check(lib.mmt_gemm_attention_bwd(C.byref(g), C.byref(a), _p(dq), dq_bs, _p(dk), _p(dv), dk_bs, dk_rs, _stream()),
               "mmt_gemm_attention_bwd")


def colsum(a, out, b=None, mask=None, accumulate=False):
    """out[n] (+)= sum_m a[m,n] * b[m,n] (b optional) over rows where mask[m,n] > 0 (mask optional)."""
    lda = _rows(a, "a")
    M, N = a.shape
    ldb = _rows(b, "b") if b is not None else 0
    ldm = _rows(mask, "mask") if mask is not None else 0
    _dev(out, "out")
    if out.numel() != N or not out.is_contiguous():
        raise MmtError("colsum: out must be a contiguous vector of N elements")
    lib = _lib.load()
    _lib.check(lib.mmt_colsum(_p(a), _p(b), _p(mask), M, N, lda, ldb, ldm, _p(out), int(bool(accumulate)), _stream()),
               "mmt_colsum")


def im2col(inp, B, C, H, W, kh, kw, stride, cols, nhwc, alpha=1.0, beta=0.0, slope=1.0):
    """cols (B*OH*OW, C*kh*kw; row stride >= that) <- patches of ``inp``: NCHW (B,C,H,W) contiguous, or rows (B*H*W, >=C
    columns) when ``nhwc``; leaky_relu(alpha*v + beta, slope) is applied on load."""
    _dev(inp, "in"), _dev(cols, "cols")
    OH, OW = (H - kh) // stride + 1, (W - kw) // stride + 1
    if tuple(cols.shape) != (B * OH * OW, C * kh * kw) or cols.stride(1) != 1:
        raise MmtError(f"im2col: cols must be ({B * OH * OW}, {C * kh * kw}) with unit column stride")
    if nhwc:
        if inp.dim() != 2 or inp.shape[0] != B * H * W or inp.shape[1] < C or inp.stride(1) != 1:
            raise MmtError("im2col: nhwc input must be (B*H*W, >=C) rows with unit column stride")
        ld = inp.stride(0)
    else:
        if tuple(inp.shape) != (B, C, H, W) or not inp.is_contiguous():
            raise MmtError("im2col: nchw input must be a contiguous (B,C,H,W) tensor")
        ld = 0
    lib = _lib.load()
    _lib.check(lib.mmt_im2col(_p(inp), B, C, H, W, kh, kw, stride, int(bool(nhwc)), ld, float(alpha), float(beta), float(slope),
                              _p(cols), cols.stride(0), _stream()), "mmt_im2col")


def col2im(dcols, B, C, H, W, kh, kw, stride, din, z=None, slope=1.0):
    """din (B*H*W, C) <- transposed im2col of dcols (B*OH*OW, C*kh*kw), times leaky_relu'(z) (z laid out like din) when given.
    Row strides may exceed the widths."""
    _dev(dcols, "dcols"), _dev(din, "din")
    OH, OW = (H - kh) // stride + 1, (W - kw) // stride + 1
    if tuple(dcols.shape) != (B * OH * OW, C * kh * kw) or dcols.stride(1) != 1:
        raise MmtError("col2im: dcols shape / column stride")
    if tuple(din.shape) != (B * H * W, C) or din.stride(1) != 1 or (
            z is not None and (tuple(z.shape) != tuple(din.shape) or z.stride() != din.stride())):
        raise MmtError("col2im: din / z must be (B*H*W, C) with unit column stride and equal row strides")
    lib = _lib.load()
    _lib.check(lib.mmt_col2im(_p(dcols), dcols.stride(0), B, C, H, W, kh, kw, stride, _p(z), float(slope), _p(din), din.stride(0),
                              _stream()), "mmt_col2im")


def ln_param_grads(dy, xhat, gw, gb, splitk=1, part_stride=0):
    """LayerNorm weight / bias gradients in one pass: gw = sum_m dy*xhat, gb = sum_m dy; with ``splitk`` > 1 the rows
    are dealt over slices and slice z lands at gw/gb + z*``part_stride`` floats (the caller sums the slices)."""
    ld = _rows(dy, "dy")
    M, N = dy.shape
    if tuple(xhat.shape) != (M, N) or _rows(xhat, "xhat") != ld:
        raise MmtError("ln_param_grads: dy and xhat must have the same shape and row stride")
    for t, n in ((gw, "gw"), (gb, "gb")):
        _dev(t, n)
        if t.numel() != N or not t.is_contiguous():
            raise MmtError(f"ln_param_grads: {n} must be a contiguous vector of N elements")
    lib = _lib.load()
    _lib.check(lib.mmt_ln_param_grads(_p(dy), _p(xhat), M, N, ld, _p(gw), _p(gb), int(splitk), int(part_stride), _stream()),
               "mmt_ln_param_grads")


# ---- group launches: many independent jobs of one kind in ONE launch ------------------------------------------------
# host table bytes -> (device copy, njobs, nblocks); the jobs' pointers are part of the key.  A launch bakes the device table's
# address into whatever HIP graph is capturing, so a table must live as long as every graph that replays it: an owner that
# captures (TrainEngine) passes its OWN dict as ``cache`` and keeps it next to its graphs -- entries of such a dict are never
# evicted, they die with the owner.  The module-level dict serves ad-hoc eager calls only (tests, tools) and is bounded.
_GROUP_TABLES = {}
_GROUP_TABLES_MAX = 64


def _group_launch(kind, jobs, plan, run, what, cache=None):
    """jobs: ctypes array of job structs.  The launch table is planned on the host and uploaded once per distinct job list
    (a later call with the same pointers and shapes -- every replay of a captured step -- finds it by its bytes).  Returns the
    device table (the caller may hold it)."""
    shared = cache is None
    if shared:
        cache = _GROUP_TABLES
    key = (kind, bytes(jobs))
    hit = cache.get(key)
    lib = _lib.load()
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            raise MmtError(f"{what}: first use of this job list inside a graph capture; run the step once eagerly first")
        nbytes, nblocks = C.c_size_t(0), C.c_int(0)
        _lib.check(plan(jobs, len(jobs), None, 0, C.byref(nbytes), C.byref(nblocks)), what + "_plan")
        host = (C.c_uint8 * nbytes.value)()
        _lib.check(plan(jobs, len(jobs), host, nbytes.value, C.byref(nbytes), C.byref(nblocks)), what + "_plan")
        table = torch.frombuffer(host, dtype=torch.uint8).clone().to(_group_device())
        if shared and len(cache) >= _GROUP_TABLES_MAX:       # ad-hoc callers come and go in a test session: keep this one bounded
            cache.pop(next(iter(cache)))
        hit = cache[key] = (table, len(jobs), nblocks.value)
    elif shared and torch.cuda.is_current_stream_capturing():
        # a graph is about to reference a table that only the bounded, shared cache keeps alive
        raise MmtError(f"{what}: group launches under graph capture need an owner's table cache (cache=...)")
    table, njobs, nblocks = hit
    _lib.check(run(C.c_void_p(table.data_ptr()), njobs, nblocks, _stream()), what)
    return table


def _group_device():
    return torch.cuda.current_device()


def gemm_group(calls, cache=None):
    """``calls``: list of keyword dicts, each the arguments of a split-K / asum ``gemm`` (the deferred weight gradients of a
    step).  One launch runs the tiles of all of them; results are those of the separate calls, bit for bit.  ``cache``: the
    owner's table dict (see ``_GROUP_TABLES``); required when the launch is captured into a graph."""
    if not calls:
        return None
    lib = _lib.load()
    jobs = (_lib.GemmArgs * len(calls))(*[_gemm_args(**kw) for kw in calls])
    return _group_launch("gemm", jobs, lib.mmt_gemm_group_plan, lib.mmt_gemm_group, "mmt_gemm_group", cache)


def ln_param_grads_group(calls, cache=None):
    """``calls``: list of keyword dicts, each the arguments of ``ln_param_grads``; one launch.  ``cache`` as in ``gemm_group``."""
    if not calls:
        return None
    lib = _lib.load()
    jobs = (_lib.LnGradJob * len(calls))()
    for J, kw in zip(jobs, calls):
        dy, xhat, gw, gb = kw["dy"], kw["xhat"], kw["gw"], kw["gb"]
        ld = _rows(dy, "dy")
        M, N = dy.shape
        if tuple(xhat.shape) != (M, N) or _rows(xhat, "xhat") != ld:
            raise MmtError("ln_param_grads: dy and xhat must have the same shape and row stride")
        for t, n in ((gw, "gw"), (gb, "gb")):
            _dev(t, n)
            if t.numel() != N or not t.is_contiguous():
                raise MmtError(f"ln_param_grads: {n} must be a contiguous vector of N elements")
        J.dy, J.xhat, J.gw, J.gb = dy.data_ptr(), xhat.data_ptr(), gw.data_ptr(), gb.data_ptr()
        J.ld, J.part_stride, J.M, J.N, J.splitk = ld, int(kw.get("part_stride", 0)), M, N, int(kw.get("splitk", 1))
    return _group_launch("ln", jobs, lib.mmt_ln_grad_group_plan, lib.mmt_ln_grad_group, "mmt_ln_grad_group", cache)


def _contig2(t, name, shape=None):
    _dev(t, name)
    if not t.is_contiguous() or (shape is not None and tuple(t.shape) != tuple(shape)):
        raise MmtError(f"{name}: expected a contiguous tensor of shape {shape}, got {tuple(t.shape)} {t.stride()}")
    return t


def layernorm_fwd(x, res, g, b, y, xhat, rstd, relu=False):
    """y = [relu](LayerNorm(x + res) * g + b), eps 1e-5; stashes xhat (M,N) and rstd (M)."""
    M, N = x.shape
    for t, n in ((x, "x"), (y, "y"), (xhat, "xhat")):
        _contig2(t, n, (M, N))
    if res is not None:
        _contig2(res, "res", (M, N))
    _contig2(rstd, "rstd", (M,))
    lib = _lib.load()
    _lib.check(lib.mmt_layernorm_fwd(_p(x), _p(res), M, N, _p(_dev(g, "g")), _p(_dev(b, "b")), int(bool(relu)), _p(y),
                                     _p(xhat), _p(rstd), _stream()), "mmt_layernorm_fwd")


def layernorm_bwd(dy, xhat, rstd, g, dx, y_relu=None, dy_masked=None, accumulate=False, drop=None):
    """dx (+)= LayerNorm backward of dy (zeroed where y_relu <= 0); the masked dy goes to dy_masked.
    ``drop=(keep, scale, dxm)``: dxm = keep*scale*dx (the gradient entering a dropped sub-layer output)."""
    M, N = dy.shape
    for t, n in ((dy, "dy"), (xhat, "xhat"), (dx, "dx")):
        _contig2(t, n, (M, N))
    _contig2(rstd, "rstd", (M,))
    if y_relu is not None:
        _contig2(y_relu, "y_relu", (M, N))
    if dy_masked is not None:
        _contig2(dy_masked, "dy_masked", (M, N))
    keep, kscale, dxm = (None, 0.0, None) if drop is None else drop
    if drop is not None:
        _contig2(dxm, "dxm", (M, N))
        if _keep(keep, "drop", (M, N)) != N:
            raise MmtError("layernorm_bwd: the keep mask must be contiguous")
    lib = _lib.load()
    _lib.check(lib.mmt_layernorm_bwd(_p(dy), _p(xhat), _p(rstd), _p(_dev(g, "g")), _p(y_relu), M, N, _p(dx),
                                     int(bool(accumulate)), _p(dy_masked), _p(keep), float(kscale), _p(dxm), _stream()),
               "mmt_layernorm_bwd")


def _bld(t, name, B, L, d):
    _dev(t, name)
    if tuple(t.shape) != (B, L, d) or t.stride(2) != 1:
        raise MmtError(f"{name}: expected a (B,L,d)=({B},{L},{d}) view with unit feature stride, got "
                       f"{tuple(t.shape)} {t.stride()}")
    return t.stride(0), t.stride(1)


def _attn_args(q, k, v, probs, pad, nhead, causal, causal_offset, scale, o=None, probs_strides=None, pdrop=None):
    B, Lq, d = q.shape
    Lk = k.shape[1]
    a = _lib.AttnArgs()
    a.q_bs, a.q_rs = _bld(q, "q", B, Lq, d)
    a.k_bs, a.k_rs = _bld(k, "k", B, Lk, d)
    a.v_bs, a.v_rs = _bld(v, "v", B, Lk, d)
    a.q, a.k, a.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    if o is not None:
        a.o_bs, a.o_rs = _bld(o, "o", B, Lq, d)
        a.o = o.data_ptr()
    _dev(probs, "probs")
    if probs_strides is None:
        if not probs.is_contiguous() or probs.numel() != B * nhead * Lq * Lk:
            raise MmtError(f"probs: expected {B * nhead * Lq * Lk} contiguous floats")
        a.p_hs, a.p_rs = Lq * Lk, Lk
    else:       # rows of a larger stash: (b,h,i) at probs[(b*H+h)*p_hs + i*p_rs : +Lk]
        a.p_hs, a.p_rs = int(probs_strides[0]), int(probs_strides[1])
        if not probs.is_contiguous() or probs.numel() < (B * nhead - 1) * a.p_hs + (Lq - 1) * a.p_rs + Lk:
            raise MmtError("probs: the strided stash is smaller than the rows addressed")
    a.probs = probs.data_ptr()
    if pdrop is not None:       # (keep bytes laid out and offset exactly like ``probs``, scale)
        keep, pscale = pdrop
        _dev(keep, "pdrop", torch.uint8)
        if not keep.is_contiguous() or keep.numel() != probs.numel():
            raise MmtError("pdrop: the keep mask must be contiguous and as large as the probs stash it shadows")
        a.pdrop, a.pdrop_scale = keep.data_ptr(), float(pscale)
    if pad is not None:
        _dev(pad, "pad", torch.uint8)
        if tuple(pad.shape) != (B, Lk) or not pad.is_contiguous():
            raise MmtError(f"pad: expected a contiguous (B,Lk)=({B},{Lk}) uint8 mask")
        a.pad = pad.data_ptr()
    if d % nhead:
        raise MmtError("attention: d_model is not divisible by nhead")
    a.B, a.H, a.Lq, a.Lk, a.dh = B, nhead, Lq, Lk, d // nhead
    a.causal, a.causal_offset, a.scale = int(bool(causal)), int(causal_offset), float(scale)
    return a


def attention_fwd(q, k, v, o, probs, nhead, scale, pad=None, causal=False, causal_offset=0, probs_strides=None, pdrop=None):
    """o = softmax(scale * q k^T + masks) v per head; q,o (B,Lq,d) and k,v (B,Lk,d) strided views; the
    probabilities (B,H,Lq,Lk) are stashed (zero at masked keys).  ``pdrop=(keep, scale)``: dropout on the probabilities
    (o uses keep*scale*softmax, the stash keeps the softmax)."""
    a = _attn_args(q, k, v, probs, pad, nhead, causal, causal_offset, scale, o=o, probs_strides=probs_strides, pdrop=pdrop)
    lib = _lib.load()
    _lib.check(lib.mmt_attention_fwd(C.byref(a), _stream()), "mmt_attention_fwd")


def attention_bwd(q, k, v, probs, d_o, dq, dk, dv, nhead, scale, probs_strides=None, pdrop=None):
    """dq (=), dk (+=), dv (+=) from d_o and the stashed probabilities; dk and dv must share strides.
    ``probs_strides`` = (stride between (b,h) slabs, stride between query rows) when ``probs`` starts inside a
    larger (B,H,Lq',Lk') stash."""
    a = _attn_args(q, k, v, probs, None, nhead, False, 0, scale, probs_strides=probs_strides, pdrop=pdrop)
    B, Lq, d = q.shape
    Lk = k.shape[1]
    do_bs, do_rs = _bld(d_o, "d_o", B, Lq, d)
    dq_bs, dq_rs = _bld(dq, "dq", B, Lq, d)
    dk_bs, dk_rs = _bld(dk, "dk", B, Lk, d)
    if _bld(dv, "dv", B, Lk, d) != (dk_bs, dk_rs):
        raise MmtError("attention_bwd: dk and dv must share strides")
    lib = _lib.load()
    _lib.check(lib.mmt_attention_bwd(C.byref(a), _p(d_o), do_bs, do_rs, _p(dq), dq_bs, dq_rs, _p(dk), _p(dv), dk_bs,
                                     dk_rs, _stream()), "mmt_attention_bwd")


def l2_loss_bwd(pos, y, dpos, loss_sum=None):
    """pos (ph,B,2) step-major, y (B,ph,2): dpos = d/dpos mean_b mean_t ||pos - y||; loss_sum[0] += that mean."""
    ph, B, _ = pos.shape
    _contig2(pos, "pos", (ph, B, 2)), _contig2(y, "y", (B, ph, 2)), _contig2(dpos, "dpos", (ph, B, 2))
    if loss_sum is not None:
        _contig2(loss_sum, "loss_sum", (1,))
    lib = _lib.load()
    _lib.check(lib.mmt_l2_loss_bwd(_p(pos), _p(y), B, ph, _p(dpos), _p(loss_sum), _stream()), "mmt_l2_loss_bwd")


def history_prepare(x_st, x0, pad):
    """x0 = nan_to_num(x_st, nan=0) (Transformer_model.py:47), pad = isnan(x_st)[:, :, 1] (model_utils.py:148-151)."""
    B, T, n = x_st.shape
    _contig2(x_st, "x_st", (B, T, n)), _contig2(x0, "x0", (B * T, n))
    _dev(pad, "pad", torch.uint8)
    if tuple(pad.shape) != (B, T) or not pad.is_contiguous():
        raise MmtError("pad: expected a contiguous (B,T) uint8 tensor")
    lib = _lib.load()
    _lib.check(lib.mmt_history_prepare(_p(x_st), B * T, n, _p(x0), _p(pad), _stream()), "mmt_history_prepare")


def dropout_mask(keep, p, rng_state, stream_id):
    """keep (uint8, contiguous, a multiple of 4 elements) <- Bernoulli(1-p) bytes from Philox4x32-10 keyed by the device
    words ``rng_state`` = int64 [seed, step] and ``stream_id``."""
    _dev(keep, "keep", torch.uint8), _dev(rng_state, "rng_state", torch.int64)
    if not keep.is_contiguous() or keep.numel() % 4 or rng_state.numel() != 2:
        raise MmtError("dropout_mask: keep must be contiguous with a multiple of 4 elements, rng_state two int64 words")
    lib = _lib.load()
    _lib.check(lib.mmt_dropout_mask(_p(keep), keep.numel(), float(p), _p(rng_state), int(stream_id), _stream()), "mmt_dropout_mask")


def rng_advance(rng_state):
    _dev(rng_state, "rng_state", torch.int64)
    lib = _lib.load()
    _lib.check(lib.mmt_rng_advance(_p(rng_state), _stream()), "mmt_rng_advance")


def dropout_apply(x, keep, scale, y):
    """y = keep ? x*scale : 0 for contiguous tensors of one shape (y may be x)."""
    _contig2(x, "x"), _contig2(y, "y", x.shape)
    _dev(keep, "keep", torch.uint8)
    if not keep.is_contiguous() or keep.numel() != x.numel():
        raise MmtError("dropout_apply: the keep mask must be contiguous and as large as x")
    lib = _lib.load()
    _lib.check(lib.mmt_dropout_apply(_p(x), _p(keep), float(scale), _p(y), x.numel(), _stream()), "mmt_dropout_apply")


def adamw_table(entries, device):
    """entries: [(param, grad, exp_avg, exp_avg_sq, lr_scale)] -> device-resident mmt_adamw_tensor table."""
    arr = (_lib.AdamwTensor * len(entries))()
    for i, (p, g, m, v, s) in enumerate(entries):
        for t, n in ((p, "param"), (g, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
            _dev(t, n)
            if not t.is_contiguous() or t.numel() != p.numel():
                raise MmtError(f"adamw: {n} of entry {i} is not a contiguous tensor of {p.numel()} elements")
        arr[i].param, arr[i].grad, arr[i].exp_avg, arr[i].exp_avg_sq = (p.data_ptr(), g.data_ptr(), m.data_ptr(),
                                                                        v.data_ptr())
        arr[i].numel, arr[i].lr_scale = p.numel(), float(s)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device), len(entries)


def adamw_step(table, ntensors, state, beta1, beta2, eps, weight_decay, clip, grad_scale, guard=None):
    """clip_grad_value_(clip) + torch.optim.AdamW over the table; state = device [step, lr].  ``guard``: 1-element device
    tensor; the update is skipped when it is not finite."""
    _dev(table, "table", torch.uint8), _contig2(state, "state", (2,))
    if guard is not None:
        _dev(guard, "guard")
    lib = _lib.load()
    _lib.check(lib.mmt_adamw_multi(_p(table), ntensors, _p(state), beta1, beta2, eps, weight_decay, clip, grad_scale,
                                   _p(guard), _stream()), "mmt_adamw_multi")


def lr_step(state, gamma, guard=None, skipped=None):
    """state[0] += 1 (optimizer step count), state[1] *= gamma (ExponentialLR stepped per batch, train.py:226-256); with a
    non-finite ``guard`` nothing but ``skipped[0] += 1``."""
    _contig2(state, "state", (2,))
    lib = _lib.load()
    _lib.check(lib.mmt_lr_step(_p(state), float(gamma), _p(guard), _p(skipped), _stream()), "mmt_lr_step")


def softmax_axis(x, y, axis):
    """y = softmax(x) over ``axis`` of a contiguous tensor (nn.Softmax)."""
    _contig2(x, "x", x.shape), _contig2(y, "y", x.shape)
    outer = 1
    for s in x.shape[:axis]:
        outer *= s
    inner = 1
    for s in x.shape[axis + 1:]:
        inner *= s
    lib = _lib.load()
    _lib.check(lib.mmt_softmax_axis(_p(x), outer, x.shape[axis], inner, _p(y), _stream()), "mmt_softmax_axis")


def cumsum_axis(x, y, axis):
    """y = cumsum(x) over ``axis`` of a contiguous tensor."""
    _contig2(x, "x", x.shape), _contig2(y, "y", x.shape)
    outer = 1
    for s in x.shape[:axis]:
        outer *= s
    inner = 1
    for s in x.shape[axis + 1:]:
        inner *= s
    lib = _lib.load()
    _lib.check(lib.mmt_cumsum_axis(_p(x), outer, x.shape[axis], inner, _p(y), _stream()), "mmt_cumsum_axis")


def l2_norm(a, b):
    """torch.norm(a - b, p=2, dim=-1) for contiguous fp32 device tensors; ``b`` has a's shape, or a's shape with ONE axis (not the
    last) of length 1 that is broadcast (model_utils.L2_norm's call shapes)."""
    if a.dim() != b.dim() or a.shape[-1] != b.shape[-1]:
        raise MmtError(f"l2_norm: shapes {tuple(a.shape)} and {tuple(b.shape)} do not line up")
    diff = [i for i in range(a.dim() - 1) if a.shape[i] != b.shape[i]]
    if len(diff) > 1 or (diff and b.shape[diff[0]] != 1):
        raise MmtError(f"l2_norm: {tuple(b.shape)} must equal {tuple(a.shape)} up to one broadcast axis of length 1")
    _contig2(a, "a", a.shape), _contig2(b, "b", b.shape)
    rep, inner = 1, 1
    if diff:
        rep = a.shape[diff[0]]
        for s_ in a.shape[diff[0] + 1:-1]:
            inner *= s_
    else:
        for s_ in a.shape[:-1]:
            inner *= s_
    rows = a.numel() // a.shape[-1]
    out = torch.empty(a.shape[:-1], dtype=torch.float32, device=a.device)
    lib = _lib.load()
    _lib.check(lib.mmt_l2_norm(_p(a), _p(b), rows, a.shape[-1], rep, inner, _p(out), _stream()), "mmt_l2_norm")
    return out


def add_inplace(y, x):
    """y += x (contiguous, same shape)."""
    _contig2(y, "y", y.shape), _contig2(x, "x", y.shape)
    lib = _lib.load()
    _lib.check(lib.mmt_add_inplace(_p(y), _p(x), y.numel(), _stream()), "mmt_add_inplace")


def patchify(img, patch, out, alpha=1.0, beta=0.0):
    """out rows (b, hy, wx) = the (p1 p2 c)-ordered pixels of patch (hy, wx) of alpha*img+beta; img (B,C,S,S) contiguous, out a
    (B*(S/P)^2, P*P*C) view with unit column stride."""
    _contig2(img, "img")
    B, Cc, S, S2 = img.shape
    _dev(out, "out")
    if S != S2 or S % patch or tuple(out.shape) != (B * (S // patch) ** 2, patch * patch * Cc) or out.stride(1) != 1:
        raise MmtError(f"patchify: img {tuple(img.shape)} / out {tuple(out.shape)} do not fit patch {patch}")
    lib = _lib.load()
    _lib.check(lib.mmt_patchify(_p(img), B, Cc, S, int(patch), float(alpha), float(beta), _p(out), out.stride(0), _stream()),
               "mmt_patchify")


def gelu_fwd(z, y):
    _contig2(z, "z"), _contig2(y, "y", z.shape)
    lib = _lib.load()
    _lib.check(lib.mmt_gelu_fwd(_p(z), _p(y), z.numel(), _stream()), "mmt_gelu_fwd")


def gelu_bwd(z, dy, dz):
    _contig2(z, "z"), _contig2(dy, "dy", z.shape), _contig2(dz, "dz", z.shape)
    lib = _lib.load()
    _lib.check(lib.mmt_gelu_bwd(_p(z), _p(dy), _p(dz), z.numel(), _stream()), "mmt_gelu_bwd")


def copy3(src, dst, accumulate=False):
    """dst (+)= src for (B,R,N) views with unit column stride; src may be an expand()ed (stride-0) view."""
    _dev(src, "src"), _dev(dst, "dst")
    if src.dim() != 3 or tuple(src.shape) != tuple(dst.shape) or src.stride(2) != 1 or dst.stride(2) != 1:
        raise MmtError(f"copy3: need equal (B,R,N) views with unit column stride, got {tuple(src.shape)} {tuple(dst.shape)}")
    B, R, N = src.shape
    lib = _lib.load()
    _lib.check(lib.mmt_copy3(_p(src), src.stride(0), src.stride(1), _p(dst), dst.stride(0), dst.stride(1), B, R, N,
                             int(bool(accumulate)), _stream()), "mmt_copy3")
