"""Training step of the hot path on HIP kernels (reference ``Transformer/train.py:218-256``
driving ``mgcvae.py:559-627``): forward with stash, reverse-time backward through the KV-cached autoregressive
decoder, deferred weight-gradient GEMMs over all (step, agent) rows, flat-bucket RCCL all-reduce, fused
clip-by-value + AdamW + ExponentialLR.

Mathematically this is the reference's autograd graph (12 full decoder passes on a growing target, BPTT through
the fed-back positions, mgcvae.py:432-446) restated on the KV-cached recurrence: hidden state j of every decoder
layer depends on target tokens <= j only, so its key/value are computed once (step j) and receive gradient from
every later step s >= j; going backwards in time, dK_j / dV_j are complete exactly when step j is reached.

Layout.  Encoder activations are agent-major rows (b*T + t), decoder activations are step-major rows (s*B + b) so
that every per-step slice is contiguous and the weight gradient of each Linear is ONE GEMM over all ph*B rows.
All buffers are allocated once per (B, T, ph); a whole step launches ~2000 small kernels and is replayed as a
HIP graph.  Parameters are used in place (natural nn.Linear layout); gradients live in one flat buffer (the
all-reduce bucket) whose slices are the per-parameter ``.grad`` views.

Dropout (Transformer_model.py:28-30,77-79,126-129; mgcvae.py:291-293; the ViT's 0.1 / 0.1): active when the modules are in
training mode and their probabilities are > 0, exactly the state torch's own modules consult.  All keep masks of a step are
drawn up front by one counter-based kernel per probability (Philox4x32-10 keyed by a device-resident {seed, step}: a replayed
graph draws fresh masks) into a uint8 pool; the forward GEMM epilogues, the attention kernels and the backward read the same
bytes.  One mask per stash position: the reference re-runs the whole decoder on the growing target with fresh masks for
every pass (mgcvae.py:432-446); here position j of the KV-cached recurrence keeps its mask for the rest of the decode (the
distribution of each activation is the same, the correlation between passes is not a parity target).
With dropout the autoregressive tokens depend on the masks: the decode reads the step's keep bytes (mmt_decoder_forward_dropout /
_stash), or, where the inference decoder kernels cannot (geometry), runs step by step on the stash kernels.  With every
probability 0 (or modules in eval mode) the dropout-free graph runs, bit for bit the one training parity is defined on.

Round 3, at the shipped widths and B <= 256: the autoregressive decode keeps the decoder's stash itself
(``TrainEngine.stash_decode``, mmt_decoder_forward_stash: no second, all-steps forward pass) and the reverse-time sweep is ONE
weight-stationary pipeline launch (``TrainEngine.sweep_pipeline``, mmt_sweep_pipe_run).  Both fall back by themselves to the
launch-per-kernel forms below (``_decoder_steps``, ``_sweep``) - other geometries, teacher forcing, B > 256, a device whose
one-launch kernels have timed out - and the tests compare each with the form it replaces.
"""
from __future__ import annotations

import math
import os

import torch

from . import _train_kernels as K
from . import ops
from ._lib import MmtError


def _t(w):
    """nn.Linear weight (N,K) -> B operand (K,N) of ``X @ W^T``."""
    return w.t()


class _Stash:
    pass


class _MlpTape:
    """Forward stash and backward of one regression-mode ``Mlp`` (MLP.py:6-30: (Linear, LayerNorm, ReLU) x n + Linear)
    applied to all ph*B rows (step-major).  ``gin[s]`` receives the gradient at the Mlp's input for step s."""

    def __init__(self, eng, mlp, ph, B):
        self.hidden = []
        i = 0
        while hasattr(mlp.layer_seq, f"lmlp_{i}"):
            seq = getattr(mlp.layer_seq, f"lmlp_{i}")
            self.hidden.append((seq[0], seq[1]))
            i += 1
        self.out = mlp.layer_seq.out_mlp
        if not isinstance(self.out, torch.nn.Linear):
            raise MmtError("only regression-mode Mlps take part in the training step")
        first = self.hidden[0][0] if self.hidden else self.out
        self.k_in, self.n_out = first.weight.shape[1], self.out.weight.shape[0]
        n, b = self.out.weight.shape[1], eng._buf
        self.ph, self.B, self.R = ph, B, ph * B
        self.hu = [b(ph, B, n) for _ in self.hidden]
        self.hxhat = [b(ph, B, n) for _ in self.hidden]
        self.hrstd = [b(ph, B) for _ in self.hidden]
        self.tmp = b(ph * B, n) if self.hidden else None
        self.ghu = [b(ph, B, n) for _ in self.hidden]        # grad wrt relu(LN(.)) output
        self.ghum = [b(ph, B, n) for _ in self.hidden]       # the same, zeroed where relu was inactive
        self.ghz = [b(ph, B, n) for _ in self.hidden]        # grad wrt the Linear output feeding the LayerNorm
        self.gin = b(ph, B, self.k_in)

    def _f2(self, t):
        return t.view(self.R, t.shape[-1])

    def forward_hidden(self, zin, rows=None):
        """zin (R, k_in) -> input of the output Linear (R, n); keeps what the backward reads.  ``rows=(r0, r1)``: zin holds
        rows r0..r1-1 only (one step of a step-by-step decode)."""
        r0, r1 = rows if rows is not None else (0, self.R)
        z = zin
        for i, (lin, ln) in enumerate(self.hidden):
            tmp = self.tmp[r0:r1]
            K.gemm(z, _t(lin.weight), tmp, bias=lin.bias)
            K.layernorm_fwd(tmp, None, ln.weight, ln.bias, self._f2(self.hu[i])[r0:r1], self._f2(self.hxhat[i])[r0:r1],
                            self.hrstd[i].view(self.R)[r0:r1], relu=True)
            z = self._f2(self.hu[i])[r0:r1]
        return z

    def forward_out(self, z, out, addend=None):
        K.gemm(z, _t(self.out.weight), out, bias=self.out.bias, addend=addend)

    def backward_step(self, s, gout, rows=None):
        """``rows=(b0, b1)``: only these rows of step s (gout holds them): the sweep runs per agent range."""
        b0, b1 = rows if rows is not None else (0, self.B)
        gin, w = gout, self.out.weight
        for i in range(len(self.hidden) - 1, -1, -1):
            lin, ln = self.hidden[i]
            K.gemm(gin, w, self.ghu[i][s][b0:b1])
            K.layernorm_bwd(self.ghu[i][s][b0:b1], self.hxhat[i][s][b0:b1], self.hrstd[i][s][b0:b1], ln.weight,
                            self.ghz[i][s][b0:b1], y_relu=self.hu[i][s][b0:b1], dy_masked=self.ghum[i][s][b0:b1])
            gin, w = self.ghz[i][s][b0:b1], lin.weight
        K.gemm(gin, w, self.gin[s][b0:b1])
        return self.gin[s][b0:b1]

    def wgrads(self, eng, zin_all, gout_all):
        z = zin_all
        for i, (lin, ln) in enumerate(self.hidden):
            eng._wgrad(self._f2(self.ghz[i]), z, lin.weight, lin.bias)
            eng._ln_grad(self._f2(self.ghum[i]), self._f2(self.hxhat[i]), ln)
            z = self._f2(self.hu[i])
        eng._wgrad(gout_all, z, self.out.weight, self.out.bias)

    def parameters(self):
        for lin, ln in self.hidden:
            yield from (lin.weight, lin.bias, ln.weight, ln.bias)
        yield from (self.out.weight, self.out.bias)


class _CnnTape:
    """CNNMapEncoder (map_encoder.py:6-36) for the training step: every conv -- and flatten + fc, a conv whose kernel covers the
    whole last feature map -- as im2col + GEMM, pre-activations kept; backward = weight-gradient GEMM, input-gradient GEMM +
    col2im (with the leaky_relu derivative) per layer.  Activations are rows (b, y, x) x channels."""
    SLOPE = 0.2                                               # F.leaky_relu(conv(x), 0.2), map_encoder.py:33

    def __init__(self, eng, convs, strides, fc, in_hw, B, alpha=1.0, beta=0.0):
        """convs: Conv2d or Conv1d modules (a Conv1d is a conv over a 1 x W map); alpha, beta: rescale of the raw input."""
        self.B, self.alpha, self.beta = B, float(alpha), float(beta)
        H, W = in_hw
        self.layers = []
        b = eng._buf
        for i, conv in enumerate(convs):
            shp = tuple(conv.weight.shape)
            Cout, Cin, kh, kw = shp if len(shp) == 4 else (shp[0], shp[1], 1, shp[2])
            self.layers.append(dict(w=conv.weight, b=conv.bias, Cin=Cin, Cout=Cout, kh=kh, kw=kw, s=int(strides[i]), H=H, W=W))
            H, W = (H - kh) // int(strides[i]) + 1, (W - kw) // int(strides[i]) + 1
        Clast = self.layers[-1]["Cout"]
        if fc.weight.shape[1] != Clast * H * W:
            raise MmtError("conv encoder: fc width does not match the conv stack")
        self.layers.append(dict(w=fc.weight, b=fc.bias, Cin=Clast, Cout=fc.weight.shape[0], kh=H, kw=W, s=1, H=H, W=W))
        self.in_shape = (B, self.layers[0]["Cin"], in_hw[0], in_hw[1])
        r4 = lambda v: (v + 3) & ~3
        for i, L in enumerate(self.layers):
            OH, OW = (L["H"] - L["kh"]) // L["s"] + 1, (L["W"] - L["kw"]) // L["s"] + 1
            rows, kc, co = B * OH * OW, L["Cin"] * L["kh"] * L["kw"], L["Cout"]
            # row strides padded to multiples of 4 floats (the pad columns stay zero): the weight-gradient GEMM over the
            # B*OH*OW rows then takes the split-K panel kernel with 16-byte loads instead of the scalar fallback
            kc4, co4 = r4(kc), r4(co)
            L.update(rows=rows, kc=kc, kc4=kc4, co4=co4, cols_buf=b(rows, kc4), z_buf=b(rows, co4), dz_buf=b(rows, co4))
            L["cols"], L["z"], L["dz"] = L["cols_buf"][:, :kc], L["z_buf"][:, :co], L["dz_buf"][:, :co]
            if i > 0:
                L["dcols"] = b(rows, kc4)[:, :kc]
            if rows >= 512:
                tiles = ((co4 + 31) // 32) * ((kc4 + 31) // 32)
                L["splitk"] = max(2, min(16, 256 // tiles))
                L["parts"] = b(L["splitk"], co4 * kc4 + co4)
                L["tot"] = b(co4 * kc4 + co4)
        self.n_out = self.layers[-1]["Cout"]

    def parameters(self):
        for L in self.layers:
            yield from (L["w"], L["b"])

    def forward(self, raw):
        """raw (B,C,H,W) -> features (B, n_out); the input rescale (map*2-1, mgcvae.py:289-290) rides in the first im2col."""
        inp, nhwc, alpha, beta, slope = raw, False, self.alpha, self.beta, 1.0
        for L in self.layers:
            K.im2col(inp, self.B, L["Cin"], L["H"], L["W"], L["kh"], L["kw"], L["s"], L["cols"], nhwc, alpha, beta, slope)
            K.gemm(L["cols"], _t(L["w"].view(L["Cout"], L["kc"])), L["z"], bias=L["b"])
            inp, nhwc, alpha, beta, slope = L["z"], True, 1.0, 0.0, self.SLOPE
        return self.layers[-1]["z"]

    def backward(self, eng, g):
        """g (B, n_out): gradient at the encoded map (any row stride)."""
        dz = g
        for i in range(len(self.layers) - 1, -1, -1):
            L = self.layers[i]
            gw, gb = eng.g(L["w"]).view(L["Cout"], L["kc"]), eng.g(L["b"])
            if "parts" in L and dz is L["dz"]:
                co4, kc4, n = L["co4"], L["kc4"], L["co4"] * L["kc4"]
                K.gemm(L["dz_buf"].t(), L["cols_buf"], L["parts"][0][:n].view(co4, kc4), splitk=L["splitk"],
                       part_stride=n + co4, asum=L["parts"][0][n:])
                K.colsum(L["parts"], L["tot"])
                gw.copy_(L["tot"][:n].view(co4, kc4)[:L["Cout"], :L["kc"]])
                gb.copy_(L["tot"][n:n + L["Cout"]])
            else:
                eng._wgrad_rows(dz, L["cols"], gw, gb)
            if i > 0:
                P = self.layers[i - 1]
                K.gemm(dz, L["w"].view(L["Cout"], L["kc"]), L["dcols"])
                K.col2im(L["dcols"], self.B, L["Cin"], L["H"], L["W"], L["kh"], L["kw"], L["s"], P["dz"], z=P["z"], slope=self.SLOPE)
                dz = P["dz"]


class _VitTape:
    """ViT map encoder for the training step (third-party vit_pytorch==0.33.2 architecture, call sites mgcvae.py:145-155,294-296;
    parity unpinned): patch rearrange + Linear, cls / position tokens, pre-norm blocks
    x += to_out(softmax(q k^T / sqrt(dh)) v), x += W2 gelu(W1 LN(x)), LayerNorm + Linear on the cls token.  Forward keeps what
    the backward needs; every weight gradient is one GEMM over all B*(n+1) token rows.  Same interface as _CnnTape."""

    def __init__(self, eng, vit, B, alpha=2.0, beta=-1.0):
        cfg = vit._cfg()
        self.eng = eng
        self.vit, self.B, self.alpha, self.beta = vit, B, float(alpha), float(beta)
        self.S, self.P, self.C = cfg["image_size"], cfg["patch_size"], cfg["channels"]
        self.n = (self.S // self.P) ** 2
        self.L = L = self.n + 1
        self.d = d = cfg["dim"]
        self.H, self.dh, self.f, self.n_out = cfg["heads"], cfg["dim_head"], cfg["mlp_dim"], cfg["num_classes"]
        self.inner = inner = self.H * self.dh
        self.scale = self.dh ** -0.5
        if L > 32 or self.dh not in (16, 32, 64):
            raise MmtError("ViT map encoder training: at most 31 patches and a head dimension of 16, 32 or 64")
        self.in_shape = (B, self.C, self.S, self.S)
        R = B * L
        b = eng._buf
        self.patches, self.pe, self.g_pe = b(B * self.n, self.P * self.P * self.C), b(B * self.n, d), b(B * self.n, d)
        depth = len(vit.transformer.layers)
        self.x = [b(B, L, d) for _ in range(2 * depth + 1)]      # block input, after attention, after feed-forward, ...
        self.blocks = []
        self.dqkv_all = b(depth, B, L, 3 * inner)                # dk / dv are accumulated into: one fill per step
        for i, (attn, ff) in enumerate(vit.transformer.layers):
            self.blocks.append(dict(
                ln1=attn.norm, wqkv=attn.fn.to_qkv.weight, wo=attn.fn.to_out[0].weight, bo=attn.fn.to_out[0].bias,
                ln2=ff.norm, w1=ff.fn.net[0].weight, b1=ff.fn.net[0].bias, w2=ff.fn.net[3].weight, b2=ff.fn.net[3].bias,
                h1=b(R, d), xhat1=b(R, d), rstd1=b(R), qkv=b(B, L, 3 * inner), o=b(B, L, inner), probs=b(B, self.H, L, L),
                h2=b(R, d), xhat2=b(R, d), rstd2=b(R), z=b(R, self.f), a=b(R, self.f), dqkv=self.dqkv_all[i]))
        self.xc, self.hc, self.xhatc, self.rstdc, self.out = b(B, d), b(B, d), b(B, d), b(B), b(B, self.n_out)
        self.gx, self.g_a, self.g_z, self.g_h, self.g_o = b(B, L, d), b(R, self.f), b(R, self.f), b(R, d), b(B, L, inner)
        self.g_hc, self.g_xc = b(B, d), b(B, d)
        self.gxm = b(R, d)                                        # keep * gx where a sub-layer output was dropped

    def _drop(self, name):
        return self.eng.drop.view.get(name)

    def parameters(self):
        return self.vit.parameters()

    def _qkv(self, t):
        i = self.inner
        return t[:, :, 0:i], t[:, :, i:2 * i], t[:, :, 2 * i:3 * i]

    def forward(self, raw):
        """raw (B,C,S,S) in [0,1] -> encoded map (B, n_out); ``map*2-1`` (mgcvae.py:289-290) rides in the patch gather."""
        v, B, L, d, n = self.vit, self.B, self.L, self.d, self.n
        R = B * L
        K.patchify(raw, self.P, self.patches, self.alpha, self.beta)
        lin = v.to_patch_embedding[1]
        K.gemm(self.patches, _t(lin.weight), self.pe, bias=lin.bias)
        x0 = self.x[0]
        K.copy3(v.pos_embedding[:, :L].expand(B, L, d), x0)
        K.copy3(v.cls_token.expand(B, 1, d), x0[:, 0:1, :], accumulate=True)
        K.copy3(self.pe.view(B, n, d), x0[:, 1:, :], accumulate=True)
        de = self._drop("vit.emb")
        if de is not None:      # self.dropout(x) after the position embedding
            K.dropout_apply(x0, de[0], de[1], x0)
        for i, T in enumerate(self.blocks):
            xin, xmid, xout = (self.x[2 * i + j].view(R, d) for j in range(3))
            K.layernorm_fwd(xin, None, T["ln1"].weight, T["ln1"].bias, T["h1"], T["xhat1"], T["rstd1"])
            K.gemm(T["h1"], _t(T["wqkv"]), T["qkv"].view(R, 3 * self.inner))
            q, k, vv = self._qkv(T["qkv"])
            K.attention_fwd(q, k, vv, T["o"], T["probs"], self.H, self.scale)
            K.gemm(T["o"].view(R, self.inner), _t(T["wo"]), xmid, bias=T["bo"], addend=xin, drop=self._drop(f"vit{i}.out"))
            K.layernorm_fwd(xmid, None, T["ln2"].weight, T["ln2"].bias, T["h2"], T["xhat2"], T["rstd2"])
            K.gemm(T["h2"], _t(T["w1"]), T["z"], bias=T["b1"])
            K.gelu_fwd(T["z"], T["a"])
            d1 = self._drop(f"vit{i}.ff1")
            if d1 is not None:
                K.dropout_apply(T["a"], d1[0], d1[1], T["a"])
            K.gemm(T["a"], _t(T["w2"]), xout, bias=T["b2"], addend=xmid, drop=self._drop(f"vit{i}.ff2"))
        K.copy3(self.x[-1][:, 0:1, :], self.xc.view(B, 1, d))                    # cls pooling
        ln, fc = v.mlp_head[0], v.mlp_head[1]
        K.layernorm_fwd(self.xc, None, ln.weight, ln.bias, self.hc, self.xhatc, self.rstdc)
        K.gemm(self.hc, _t(fc.weight), self.out, bias=fc.bias)
        return self.out

    def backward(self, eng, g):
        """g (B, n_out): gradient at the encoded map (any row stride)."""
        v, B, L, d, n = self.vit, self.B, self.L, self.d, self.n
        R = B * L
        ln, fc = v.mlp_head[0], v.mlp_head[1]
        eng._wgrad_rows(g, self.hc, eng.g(fc.weight), eng.g(fc.bias))
        K.gemm(g, fc.weight, self.g_hc)
        eng._ln_grad(self.g_hc, self.xhatc, ln)
        K.layernorm_bwd(self.g_hc, self.xhatc, self.rstdc, ln.weight, self.g_xc)
        self.gx.zero_()
        self.dqkv_all.zero_()
        K.copy3(self.g_xc.view(B, 1, d), self.gx[:, 0:1, :])
        gx = self.gx.view(R, d)
        for i in range(len(self.blocks) - 1, -1, -1):
            T = self.blocks[i]
            # x_out = x_mid + drop(W2 drop(gelu(W1 LN2(x_mid) + b1)) + b2)
            d2, d1, do = self._drop(f"vit{i}.ff2"), self._drop(f"vit{i}.ff1"), self._drop(f"vit{i}.out")
            gy = gx
            if d2 is not None:
                K.dropout_apply(gx, d2[0], d2[1], self.gxm)
                gy = self.gxm
            eng._wgrad_rows(gy, T["a"], eng.g(T["w2"]), eng.g(T["b2"]))
            K.gemm(gy, T["w2"], self.g_a)
            if d1 is not None:
                K.dropout_apply(self.g_a, d1[0], d1[1], self.g_a)
            K.gelu_bwd(T["z"], self.g_a, self.g_z)
            eng._wgrad_rows(self.g_z, T["h2"], eng.g(T["w1"]), eng.g(T["b1"]))
            K.gemm(self.g_z, T["w1"], self.g_h)
            eng._ln_grad(self.g_h, T["xhat2"], T["ln2"])
            K.layernorm_bwd(self.g_h, T["xhat2"], T["rstd2"], T["ln2"].weight, gx, accumulate=True)
            # x_mid = x_in + drop(Wo attention(LN1(x_in) Wqkv^T) + bo)
            gy = gx
            if do is not None:
                K.dropout_apply(gx, do[0], do[1], self.gxm)
                gy = self.gxm
            eng._wgrad_rows(gy, T["o"].view(R, self.inner), eng.g(T["wo"]), eng.g(T["bo"]))
            K.gemm(gy, T["wo"], self.g_o.view(R, self.inner))
            q, k, vv = self._qkv(T["qkv"])
            dq, dk, dv = self._qkv(T["dqkv"])
            K.attention_bwd(q, k, vv, T["probs"], self.g_o, dq, dk, dv, self.H, self.scale)
            dqkv = T["dqkv"].view(R, 3 * self.inner)
            eng._wgrad_rows(dqkv, T["h1"], eng.g(T["wqkv"]), None)
            K.gemm(dqkv, T["wqkv"], self.g_h)
            eng._ln_grad(self.g_h, T["xhat1"], T["ln1"])
            K.layernorm_bwd(self.g_h, T["xhat1"], T["rstd1"], T["ln1"].weight, gx, accumulate=True)
        # tokens = drop(cat[cls, patches W^T + b] + pos)
        de = self._drop("vit.emb")
        if de is not None:
            K.dropout_apply(self.gx, de[0], de[1], self.gx)
        gpos = eng.g(v.pos_embedding)
        K.colsum(self.gx.view(B, L * d), gpos.view(L * d))
        K.copy3(gpos.view(1, L, d)[:, 0:1, :], eng.g(v.cls_token).view(1, 1, d))
        K.copy3(self.gx[:, 1:, :], self.g_pe.view(B, n, d))
        lin = v.to_patch_embedding[1]
        eng._wgrad_rows(self.g_pe, self.patches, eng.g(lin.weight), eng.g(lin.bias))


class _DropPlan:
    """Keep masks of one training step: a site is (name, shape of the tensor it masks, probability).  Sites with p = 0 get no
    mask (their kernels then run the dropout-free code path).  The pool is one uint8 buffer, sites of equal probability
    contiguous, so that ``generate`` is one kernel launch per distinct probability."""

    def __init__(self, device, seed=0):
        self.device = device
        self.want = []                 # (name, shape, p)
        self.view = {}                 # name -> (uint8 view, 1/(1-p)) | None
        self.regions = []              # (p, lo, hi) in pool bytes
        self.pool = None
        self.rng = torch.tensor([int(seed), 0], dtype=torch.int64).to(device)

    def add(self, name, shape, p):
        if name in self.view or any(n == name for n, _, _ in self.want):
            raise MmtError(f"dropout site {name} registered twice")
        if not (0.0 <= float(p) < 1.0):
            raise MmtError(f"dropout probability {p} of {name} is outside [0, 1)")
        if float(p) > 0.0:
            self.want.append((name, tuple(int(v) for v in shape), float(p)))
        else:
            self.view[name] = None

    def build(self):
        total, by_p = 0, {}
        for name, shape, p in self.want:
            by_p.setdefault(p, []).append((name, shape))
        layout = []
        for p, sites in by_p.items():
            lo = total
            for name, shape in sites:
                n = 1
                for v in shape:
                    n *= v
                layout.append((name, shape, p, total, n))
                total += (n + 15) & ~15
            self.regions.append((p, lo, total))
        self.pool = torch.zeros(max(total, 16), dtype=torch.uint8, device=self.device)
        for name, shape, p, off, n in layout:
            self.view[name] = (self.pool[off:off + n].view(shape), 1.0 / (1.0 - p))

    @property
    def active(self):
        return bool(self.regions)

    def get(self, name, index=None):
        """(keep view, scale) of a site, optionally of one slice along its first axis; None when the site is not dropped."""
        v = self.view[name]
        if v is None or index is None:
            return v
        return v[0][index], v[1]

    def seed(self, seed, step=0):
        self.rng.copy_(torch.tensor([int(seed), int(step)], dtype=torch.int64))

    def generate(self):
        if self.frozen:
            return
        for i, (p, lo, hi) in enumerate(self.regions):
            K.dropout_mask(self.pool[lo:hi], p, self.rng, i)
        K.rng_advance(self.rng)

    frozen = False

    def inject(self, keep):
        """Use GIVEN keep masks {site: bool / uint8 tensor shaped like the site} from now on instead of drawing them (tests: the
        masks the real reference was run with, tests/golden/*_dropout.npz).  Every active site must be given, nothing else."""
        active = {n for n, v in self.view.items() if v is not None}
        if set(keep) != active:
            raise MmtError(f"inject: masks for {sorted(set(keep) ^ active)} missing or unknown")
        for n, m in keep.items():
            dst = self.view[n][0]
            m = torch.as_tensor(m)
            if m.numel() != dst.numel():
                raise MmtError(f"inject: site {n} has {dst.numel()} elements, mask has {m.numel()}")
            dst.copy_((m.reshape(dst.shape) != 0).to(torch.uint8))
        self.frozen = True

    def masks(self):
        """{site: bool tensor} of the masks the last ``generate`` drew (tests inject them into the oracle)."""
        return {n: (v[0] != 0) for n, v in self.view.items() if v is not None}


class TrainEngine:
    """Forward-with-stash + backward for one (B, T, ph) shape of the Basic, Map_encoding_CNN, Map_encoding_ViT or
    Lane_encoding_CNN configuration.  ``dropout``: None = as the modules say (training mode and p > 0), False = never."""

    def __init__(self, model, B, T, ph, device=None, teacher_forced=None, dropout=None, seed=0):
        hp = model.hyperparams
        # non_at_dec (mgcvae.py:310-373): tokens come from the labels, positions are the running sum of the head
        # outputs and are NOT fed back, so no gradient flows through the tokens
        self.teacher_forced = (not hp.get("autoregressive", True)) if teacher_forced is None else bool(teacher_forced)
        self.vit_mode = bool(hp["map_vit_encoding"])
        self.map_mode = bool(hp["map_cnn_encoding"]) or self.vit_mode
        self.lane_mode = bool(hp["lane_cnn_encoding"])
        if (self.map_mode or self.lane_mode) and self.teacher_forced:
            raise MmtError("non_at_dec is only defined for the Basic configuration (mgcvae.py:365-367 does not run)")
        self.cnn = model._m("map_encoder") if self.map_mode else None
        self.fuse = model._m("fusion/hist_map") if self.map_mode else None
        # lane configuration (mgcvae.py:395-424): memory <- fusion/lane_hist(cat[memory, lane feature]); the reference feeds
        # lane 0's feature to all three decodes (:404), which are therefore identical: one decode, loss = its loss (:603-612)
        self.lane_enc = model._m("lane_encoder") if self.lane_mode else None
        self.memfuse = model._m("fusion/lane_hist") if self.lane_mode else None
        if self.lane_mode:
            self.n_lane, self.n_samp = int(hp["max_lane_num"]), int(hp["sample_num"])
            self.lane_emb = int(hp["lane_encoder"][model.node_type]["embedding_size"])
        self.enc = model._m("node_history_encoder")
        self.dec = model._m("decoder/transformer_decoder")
        self.head = model._m("decoder/MLP")
        self.B, self.T, self.ph = int(B), int(T), int(ph)
        if not (1 <= self.T <= 32 and 1 <= self.ph <= 32):
            raise MmtError("training kernels support 1 <= T, ph <= 32")
        self.device = device if device is not None else self.enc.input_fc.weight.device
        self.d = self.enc.input_fc.weight.shape[0]
        self.n_in = self.enc.input_fc.weight.shape[1]
        self.H = self.enc._nhead()
        self.dh = self.d // self.H
        if self.dh > 64:
            raise MmtError("head dimension > 64 is not supported by the attention kernels")
        self.scale = 1.0 / math.sqrt(self.dh)
        self.Le = len(self.enc.transformer_encoder.layers)
        self.Ld = len(self.dec.transformer_decoder.layers)
        self.f_enc = self.enc.transformer_encoder.layers[0].linear1.weight.shape[0]
        self.f_dec = self.dec.transformer_decoder.layers[0].linear1.weight.shape[0]
        self.n_pos = self.dec.input_fc.weight.shape[1]
        self.n_out = self.dec.output_fc.weight.shape[0]
        self._pdec = None
        self._group_cache = {}          # device launch tables of the group launches (K._group_launch), owned by this engine
        self.hp = hp
        self.use_dropout = bool(self.enc.training) if dropout is None else bool(dropout)
        self.drop = _DropPlan(self.device, seed)
        self._collect_params(model)
        self._plan_dropout()
        self._alloc()

    # ------------------------------------------------------------------ dropout sites
    def _plan_dropout(self):
        """One site per nn.Dropout / F.dropout / attention-probability dropout the reference's modules apply in training mode,
        shaped like the stash tensor it masks (decoder sites step-major, like the stash)."""
        B, T, ph, d, H = self.B, self.T, self.ph, self.d, self.H
        Re, on = B * T, self.use_dropout
        P = lambda v: float(v) if on else 0.0
        add = self.drop.add
        add("enc.pe", (Re, d), P(self.enc.pos_encoder.dropout.p))
        for l, lay in enumerate(self.enc.transformer_encoder.layers):
            add(f"enc{l}.attn", (B * H * T * T,), P(lay.self_attn.dropout))
            add(f"enc{l}.drop1", (Re, d), P(lay.dropout1.p))
            add(f"enc{l}.ff", (Re, self.f_enc), P(lay.dropout.p))
            add(f"enc{l}.drop2", (Re, d), P(lay.dropout2.p))
        add("dec.pe", (ph, B, d), P(self.dec.pos_encoder.dropout.p))
        for l, lay in enumerate(self.dec.transformer_decoder.layers):
            add(f"dec{l}.attn1", (B * H * ph * ph,), P(lay.self_attn.dropout))
            add(f"dec{l}.drop1", (ph, B, d), P(lay.dropout1.p))
            add(f"dec{l}.attn2", (B * H * ph * T,), P(lay.multihead_attn.dropout))
            add(f"dec{l}.drop2", (ph, B, d), P(lay.dropout2.p))
            add(f"dec{l}.ff", (ph, B, self.f_dec), P(lay.dropout.p))
            add(f"dec{l}.drop3", (ph, B, d), P(lay.dropout3.p))
        if self.map_mode and not self.vit_mode:     # F.dropout(encoded_map, p, training=TRAIN), mgcvae.py:291-293
            add("map.feature", (B, int(self.hp["map_encoder"]["cnn_param"]["output_size"])),
                P(self.hp["map_encoder"]["cnn_param"]["dropout"]))
        if self.vit_mode:
            v = self.cnn
            cfg = v._cfg()
            Lv, dv = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1, cfg["dim"]
            von = on and v.training
            add("vit.emb", (B, Lv, dv), float(getattr(v, "dropout", torch.nn.Dropout(0.0)).p) if von else 0.0)
            for i, (attn, ff) in enumerate(v.transformer.layers):
                add(f"vit{i}.out", (B * Lv, dv), float(attn.fn.to_out[1].p) if von else 0.0)
                add(f"vit{i}.ff1", (B * Lv, cfg["mlp_dim"]), float(ff.fn.net[2].p) if von else 0.0)
                add(f"vit{i}.ff2", (B * Lv, dv), float(ff.fn.net[4].p) if von else 0.0)
        self.drop.build()

    def seed(self, seed, step=0):
        """Seed of the dropout masks (the masks of step n are a function of (seed, n) only)."""
        self.drop.seed(seed, step)

    # ------------------------------------------------------------------ parameters / gradients
    def _collect_params(self, model):
        """(registrar key + parameter name, parameter) for every parameter that receives a gradient in the Basic
        graph; Encoder.output_fc and Trajectory_Decoder.mix_fc never do and are skipped by the
        optimizer exactly like torch skips ``grad is None``."""
        nt = model.node_type
        named = []
        mods = [(nt + "/node_history_encoder", self.enc), (nt + "/decoder/transformer_decoder", self.dec),
                (nt + "/decoder/MLP", self.head)]
        if self.map_mode:
            mods += [(nt + "/map_encoder", self.cnn), (nt + "/fusion/hist_map", self.fuse)]
        if self.lane_mode:
            mods += [(nt + "/lane_encoder", self.lane_enc), (nt + "/fusion/lane_hist", self.memfuse)]
        for key, mod in mods:
            for pn, p in mod.named_parameters():
                if key.endswith("node_history_encoder") and pn.startswith("output_fc."):
                    continue
                if pn.startswith("mix_fc."):
                    continue
                named.append((f"{key}.{pn}", p))
        self.named_params = named
        total = 0
        offs = []
        enc_key, late_keys = nt + "/node_history_encoder.", (nt + "/lane_encoder.", nt + "/fusion/lane_hist.")
        enc_end, late_start = 0, None
        for name, p in named:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise MmtError("parameters must be contiguous fp32")
            if late_start is None and name.startswith(late_keys):
                late_start = total
            offs.append(total)
            total += (p.numel() + 63) & ~63
            if name.startswith(enc_key):
                enc_end = total
        # backward() finishes the gradients in two stretches: decoder / head / map encoder / fusion first (the reverse-time sweep
        # and the deferred weight-gradient GEMMs), then the history encoder (and the lane encoder / fusion/lane_hist, whose
        # gradients come out of the memory's).  ``early_range`` can be all-reduced while the second stretch still runs.
        late_start = total if late_start is None else late_start
        self.early_range = (enc_end, late_start)
        self.late_ranges = [r for r in ((0, enc_end), (late_start, total + 64)) if r[1] > r[0]]      # (+ the guard slot, below)
        # [0] is the gradient bucket (what the all-reduce and the optimizer see); [1..SPLITK-1] receive the other K-slices of
        # the split weight-gradient GEMMs at the same offsets and are summed into [0] at the end of backward()
        self.SPLITK = 4
        # one more slot behind the gradients: the step's loss rides in the all-reduce bucket, so that every data-parallel rank
        # sees the same (summed) value and skips the optimizer together when a rank's step went NaN (Trainer._opt's guard)
        self.guard_at = total
        total += 64
        self.grad_total = total
        self.flat_all = torch.zeros(self.SPLITK * total, dtype=torch.float32, device=self.device)
        self.flat_grad = self.flat_all[:total]
        self.guard = self.flat_grad[self.guard_at:self.guard_at + 1]
        self.grad_of = {}
        self.grad_by_name = {}
        for (name, p), o in zip(named, offs):
            g = self.flat_grad[o:o + p.numel()].view(p.shape)
            self.grad_of[id(p)] = g
            self.grad_by_name[name] = g

    def g(self, p):
        return self.grad_of[id(p)]

    def _buf(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def _alloc(self):
        B, T, ph, d, H = self.B, self.T, self.ph, self.d, self.H
        Re, S = B * T, _Stash()
        b = self._buf
        # static inputs
        self.x_st = b(B, T, self.n_in)
        self.y_st = b(B, ph, self.n_pos)
        self.loss = b(1)
        # encoder
        S.x0, S.pad = b(Re, self.n_in), b(B, T, dtype=torch.uint8)
        S.pe_enc = self.enc.pos_encoder.pe[:T].to(self.device).repeat(B, 1).contiguous()
        S.pe_dec = self.dec.pos_encoder.pe[:ph].to(self.device).repeat_interleave(B, dim=0).contiguous()   # row s*B+b
        fe, fd = self.f_enc, self.f_dec
        S.eh = [b(Re, d) for _ in range(self.Le + 1)]
        S.eqkv = [b(Re, 3 * d) for _ in range(self.Le)]
        S.eprobs = [b(B * H * T * T) for _ in range(self.Le)]
        S.eatt, S.eh1 = [b(Re, d) for _ in range(self.Le)], [b(Re, d) for _ in range(self.Le)]
        S.exhat1, S.exhat2 = [b(Re, d) for _ in range(self.Le)], [b(Re, d) for _ in range(self.Le)]
        S.erstd1, S.erstd2 = [b(Re) for _ in range(self.Le)], [b(Re) for _ in range(self.Le)]
        S.ef1 = [b(Re, fe) for _ in range(self.Le)]
        S.etmp = b(Re, d)
        # encoder gradients (per layer, so the weight-gradient GEMMs can run after the sweep)
        S.geh = [b(Re, d) for _ in range(self.Le + 1)]
        S.gef2, S.gef1 = [b(Re, d) for _ in range(self.Le)], [b(Re, fe) for _ in range(self.Le)]
        S.geh1, S.geo1 = [b(Re, d) for _ in range(self.Le)], [b(Re, d) for _ in range(self.Le)]
        # gradient entering a sub-layer whose output was dropped before the residual add (= the plain one without dropout)
        D = self.drop.get
        S.gef2m = [b(Re, d) if D(f"enc{l}.drop2") else S.gef2[l] for l in range(self.Le)]
        S.geo1m = [b(Re, d) if D(f"enc{l}.drop1") else S.geo1[l] for l in range(self.Le)]
        S.geatt = [b(Re, d) for _ in range(self.Le)]
        # the buffers that the attention backward kernels accumulate into (dK, dV += ...) share one pool: one fill per step
        L_ = self.Ld
        n_geqkv, n_gqkv, n_gmkv = Re * 3 * d, ph * B * 3 * d, Re * 2 * d
        S.acc_pool = b(self.Le * n_geqkv + L_ * (n_gqkv + n_gmkv))
        o = 0
        S.geqkv = []
        for _ in range(self.Le):
            S.geqkv.append(S.acc_pool[o:o + n_geqkv].view(Re, 3 * d))
            o += n_geqkv
        # decoder activations, step-major
        L = self.Ld
        S.pos = b(ph + 1, B, self.n_pos)
        S.tok = b(ph, B, self.n_pos) if self.teacher_forced else S.pos[:ph]      # decoder input tokens, step-major
        # Where the pipeline decode can keep the stash itself (mmt_decoder_forward_stash), the decoder's activations are views
        # into ONE buffer laid out as the library says; otherwise separate buffers, filled by the all-steps pass (_decoder_steps).
        lay = self._stash_layout()
        if lay is not None:
            S.dec_stash = b(lay["total_floats"])
            R_ = ph * B

            def view(l, name, n, shape):
                o = (lay["tail"] if l is None else l * lay["layer_floats"]) + lay[name]
                return S.dec_stash[o:o + n].view(*shape)
            S.x = [view(l, "x", R_ * d, (ph, B, d)) for l in range(L)] + [view(None, "t_x", R_ * d, (ph, B, d))]
            S.qkv = [view(l, "qkv", R_ * 3 * d, (ph, B, 3 * d)) for l in range(L)]
            S.p1 = [view(l, "p1", B * H * ph * ph, (B * H * ph * ph,)) for l in range(L)]
            S.p2 = [view(l, "p2", B * H * ph * T, (B * H * ph * T,)) for l in range(L)]
            for n in ("a1", "x1", "cq", "a2", "x2", "xhat1", "xhat2", "xhat3"):
                setattr(S, n, [view(l, n, R_ * d, (ph, B, d)) for l in range(L)])
            for n in ("rstd1", "rstd2", "rstd3"):
                setattr(S, n, [view(l, n, R_, (ph, B)) for l in range(L)])
            S.f1 = [view(l, "f1", R_ * fd, (ph, B, fd)) for l in range(L)]
            self._stash_view = view
        else:
            S.dec_stash = None
            S.x = [b(ph, B, d) for _ in range(L + 1)]
            S.qkv = [b(ph, B, 3 * d) for _ in range(L)]
            S.p1 = [b(B * H * ph * ph) for _ in range(L)]      # (B,H,ph,ph) causal self-attention probabilities
            S.p2 = [b(B * H * ph * T) for _ in range(L)]       # (B,H,ph,T)
            for n in ("a1", "x1", "cq", "a2", "x2", "xhat1", "xhat2", "xhat3"):
                setattr(S, n, [b(ph, B, d) for _ in range(L)])
            for n in ("rstd1", "rstd2", "rstd3"):
                setattr(S, n, [b(ph, B) for _ in range(L)])
            S.f1 = [b(ph, B, fd) for _ in range(L)]
        # Cross-attention K/V of the memory, per layer.  The decoder kernels compute them in their prologue, into the head of their
        # workspace ([layer][b*T + t][k | v]): with an engine-owned workspace the stash IS that region (six GEMMs less per step);
        # the paths that do not run those kernels (teacher forcing, the step-by-step decode) fill the same views by GEMM.
        self._dec_ws = None
        if self.alias_memory_kv and not self.teacher_forced and torch.device(self.device).type == "cuda":
            self._dec_ws = b(ops.decoder_workspace_floats(self._packed_decoder(), B, T, ph) + 64)
            self.__dict__.pop("_pdec_fresh", None)
            S.mkv = [self._dec_ws[l * Re * 2 * d:(l + 1) * Re * 2 * d].view(Re, 2 * d) for l in range(L)]
        else:
            S.mkv = [b(Re, 2 * d) for _ in range(L)]
        S.tmp = b(ph * B, d)
        no = self.n_out
        self.head_tape = _MlpTape(self, self.head, ph, B)
        if self.map_mode:
            # head input = fusion/hist_map(cat[output_fc(x), encoded_map])   (mgcvae.py:438-443)
            if self.vit_mode:
                self.cnn_tape = _VitTape(self, self.cnn, B, 2.0, -1.0)
            else:
                self.cnn_tape = _CnnTape(self, self.cnn.convs, self.cnn._strides(), self.cnn.fc, self.cnn.in_hw, B, 2.0, -1.0)
            self.fuse_tape = _MlpTape(self, self.fuse, ph, B)
            md = self.cnn_tape.n_out
            if self.fuse_tape.k_in != no + md or self.fuse_tape.n_out != self.head_tape.k_in:
                raise MmtError("fusion/hist_map width does not match output_fc + map encoder")
            if S.dec_stash is not None:     # cat[output_fc, map feature], the fusion's output and the head tape live in the tail block
                R_, ht = ph * B, self.head_tape
                if lay["ho_ld"] != no + md or self.fuse_tape.hidden or self.fuse_tape.n_out != no:
                    raise MmtError("decoder stash layout does not match fusion/hist_map")
                S.hcat = self._stash_view(None, "t_ho", R_ * (no + md), (ph, B, no + md))
                S.u = self._stash_view(None, "t_u", R_ * no, (ph, B, no))
                ht.hu[0] = self._stash_view(None, "t_hu", R_ * 32, (ph, B, 32))
                ht.hxhat[0] = self._stash_view(None, "t_hxhat", R_ * 32, (ph, B, 32))
                ht.hrstd[0] = self._stash_view(None, "t_hrstd", R_, (ph, B))
            else:
                S.hcat = b(ph, B, no + md)
                S.u = b(ph, B, self.fuse_tape.n_out)
            S.ho = S.hcat[:, :, :no]                       # output_fc writes straight into its columns of the concatenation
            S.gmap = b(B * (no + md))
            S.gmap_c = b(B, md)
            if self.drop.view.get("map.feature") is not None:
                S.enc_map_d = b(B, md)
            self.maps = b(*self.cnn_tape.in_shape)
        else:
            if self.head_tape.k_in != no:
                raise MmtError("decoder/MLP width does not match output_fc")
            if S.dec_stash is not None:     # output_fc's rows and the head's hidden layer live in the stash's tail block
                R_, ht = ph * B, self.head_tape
                S.ho = self._stash_view(None, "t_ho", R_ * no, (ph, B, no))
                ht.hu[0] = self._stash_view(None, "t_hu", R_ * 32, (ph, B, 32))
                ht.hxhat[0] = self._stash_view(None, "t_hxhat", R_ * 32, (ph, B, 32))
                ht.hrstd[0] = self._stash_view(None, "t_hrstd", R_, (ph, B))
            else:
                S.ho = b(ph, B, no)
        S.delta = b(ph, B, self.n_pos)
        S.pred = b(B, ph, self.n_pos)                      # autoregressive decode, agent-major (kernel output)
        # decoder gradients, step-major (kept for the deferred weight-gradient GEMMs)
        S.G = b(ph + 1, B, self.n_pos)
        S.gx = [b(ph, B, d) for _ in range(L + 1)]
        for n in ("gf2", "gx2", "go2", "ga2", "gcq", "gx1", "go1", "ga1"):
            setattr(S, n, [b(ph, B, d) for _ in range(L)])
        S.gf1 = [b(ph, B, fd) for _ in range(L)]
        S.gf2m = [b(ph, B, d) if D(f"dec{l}.drop3") else S.gf2[l] for l in range(L)]
        S.go2m = [b(ph, B, d) if D(f"dec{l}.drop2") else S.go2[l] for l in range(L)]
        S.go1m = [b(ph, B, d) if D(f"dec{l}.drop1") else S.go1[l] for l in range(L)]
        S.gqkv, S.gmkv = [], []
        for _ in range(L):
            S.gqkv.append(S.acc_pool[o:o + n_gqkv].view(ph, B, 3 * d))
            o += n_gqkv
            S.gmkv.append(S.acc_pool[o:o + n_gmkv].view(Re, 2 * d))
            o += n_gmkv
        if self.lane_mode:
            if T < self.n_lane:
                raise MmtError("lane decoding needs history length >= max_lane_num (mgcvae.py:404)")
            nl = self.n_lane
            stride = getattr(self.lane_enc, "stride", None) or self.lane_enc.convs[0].stride[0]
            self.lane_tape = _CnnTape(self, self.lane_enc.convs, [stride], self.lane_enc.fc, (1, self.n_samp), B * nl)
            self.memfuse_tape = _MlpTape(self, self.memfuse, 1, Re)
            lf = self.lane_tape.n_out
            if self.memfuse_tape.k_in != d + lf or self.memfuse_tape.n_out != d:
                raise MmtError("fusion/lane_hist width does not match memory + lane feature")
            self.lanes = b(B, nl, self.n_samp, self.lane_emb)
            S.memcat, S.memf, S.gmemf = b(Re, d + lf), b(Re, d), b(Re, d)
            S.glane = b(B, nl, lf)
        # gradient at output_fc's output: the head's (or the fusion's) input gradient, first n_out columns
        S.gho = (self.fuse_tape.gin[:, :, :no] if self.map_mode else self.head_tape.gin)
        self.s = S

    # ------------------------------------------------------------------ forward
    def set_batch(self, x_st, y_st, maps=None, lanes=None):
        if tuple(x_st.shape) != tuple(self.x_st.shape) or tuple(y_st.shape) != tuple(self.y_st.shape):
            raise MmtError(f"batch shape {tuple(x_st.shape)} / {tuple(y_st.shape)} does not match the engine's "
                           f"{tuple(self.x_st.shape)} / {tuple(self.y_st.shape)}")
        self.x_st.copy_(x_st)
        self.y_st.copy_(y_st)
        if self.map_mode:
            if maps is None or tuple(maps.shape) != tuple(self.maps.shape):
                raise MmtError(f"Map_encoding_CNN training needs maps of shape {tuple(self.maps.shape)}")
            self.maps.copy_(maps)
        if self.lane_mode:
            if lanes is None or lanes.numel() != self.lanes.numel():
                raise MmtError(f"Lane_encoding_CNN training needs lanes of shape {tuple(self.lanes.shape)}")
            self.lanes.copy_(lanes.reshape(self.lanes.shape))

    def forward(self):
        """Encoder -> memory K/V -> ph decoder steps with position feedback; everything the backward needs is kept."""
        S, B, T, d, H, ph = self.s, self.B, self.T, self.d, self.H, self.ph
        enc, dec = self.enc, self.dec
        D = self.drop.get
        if self.drop.active:
            self.drop.generate()
        self._prepack()
        K.history_prepare(self.x_st, S.x0, S.pad)
        K.gemm(S.x0, _t(enc.input_fc.weight), S.eh[0], bias=enc.input_fc.bias, addend=S.pe_enc, drop=D("enc.pe"),
               drop_after_add=True)
        for l, lay in enumerate(enc.transformer_encoder.layers):
            sa = lay.self_attn
            K.gemm(S.eh[l], _t(sa.in_proj_weight), S.eqkv[l], bias=sa.in_proj_bias)
            qkv = S.eqkv[l].view(B, T, 3 * d)
            K.attention_fwd(qkv[:, :, 0:d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], S.eatt[l].view(B, T, d), S.eprobs[l],
                            H, self.scale, pad=S.pad, causal=True, pdrop=D(f"enc{l}.attn"))
            K.gemm(S.eatt[l], _t(sa.out_proj.weight), S.etmp, bias=sa.out_proj.bias, addend=S.eh[l], drop=D(f"enc{l}.drop1"))
            K.layernorm_fwd(S.etmp, None, lay.norm1.weight, lay.norm1.bias, S.eh1[l], S.exhat1[l], S.erstd1[l])
            K.gemm(S.eh1[l], _t(lay.linear1.weight), S.ef1[l], bias=lay.linear1.bias, relu=True, drop=D(f"enc{l}.ff"))
            K.gemm(S.ef1[l], _t(lay.linear2.weight), S.etmp, bias=lay.linear2.bias, addend=S.eh1[l], drop=D(f"enc{l}.drop2"))
            K.layernorm_fwd(S.etmp, None, lay.norm2.weight, lay.norm2.bias, S.eh[l + 1], S.exhat2[l], S.erstd2[l])
        memory = S.eh[self.Le]
        if self.lane_mode:
            # raw reinterpretation of (B,3,10,2) as (B*3, 2, 10) (mgcvae.py:301-302), then lane 0's feature for every row
            nl = self.n_lane
            S.enc_lane = self.lane_tape.forward(self.lanes.view(B * nl, self.lane_emb, 1, self.n_samp)).view(B, nl, -1)
            S.memcat[:, :d].copy_(memory)
            S.memcat.view(B, T, -1)[:, :, d:].copy_(S.enc_lane[:, 0:1, :].expand(B, T, -1))
            self.memfuse_tape.forward_out(self.memfuse_tape.forward_hidden(S.memcat), S.memf)
            memory = S.memf
        layers = dec.transformer_decoder.layers
        # (the decoder kernels leave the memory's K/V in S.mkv themselves: see _alloc)
        kernels_decode = self._dec_ws is not None and not self.teacher_forced and \
            not (self.drop.active and not self._ar_decode_takes_masks())
        if not kernels_decode:
            for l, lay in enumerate(layers):
                ca = lay.multihead_attn
                K.gemm(memory, _t(ca.in_proj_weight[d:]), S.mkv[l], bias=ca.in_proj_bias[d:])
        if self.teacher_forced:     # tgt = cat[0, labels_st[:, :-1]], positions start from 0 (mgcvae.py:334-335,370)
            S.pos[0].zero_()
            S.tok[0].zero_()
            if ph > 1:
                S.tok[1:].copy_(self.y_st[:, :ph - 1].permute(1, 0, 2))
        else:
            S.pos[0].copy_(self.x_st[:, -1, 0:self.n_pos])      # start token = standardised current position
        if self.map_mode:
            S.enc_map = self.cnn_tape.forward(self.maps)
            S.enc_map_used = S.enc_map
            dm = self.drop.view.get("map.feature")
            if dm is not None:      # F.dropout(encoded_map, 0.5, training=True), mgcvae.py:291-293
                S.enc_map_d.copy_(S.enc_map)
                K.dropout_apply(S.enc_map_d, dm[0], dm[1], S.enc_map_d)
                S.enc_map_used = S.enc_map_d
        self._join_packs()
        if self.teacher_forced:     # positions = running sum of the head outputs (mgcvae.py:352,370), from pos[0] = 0
            self._decoder_steps(0, ph)
            R = ph * B
            self.head_tape.forward_out(self._head_z, S.delta.view(R, self.n_pos))
            K.cumsum_axis(S.delta.view(1, ph, B * self.n_pos), S.pos[1:].view(1, ph, B * self.n_pos), 1)
        elif self.drop.active and not self._ar_decode_takes_masks():
            # With dropout the fed-back positions depend on the masks.  Where the inference decoder kernels cannot read them (not
            # the 4-agent-cluster geometry) the decode runs step by step on the stash kernels (the KV-cached recurrence, one
            # query per agent against the keys of steps 0..s), which also IS the stash pass.
            for s_ in range(ph):
                self._decoder_steps(s_, s_ + 1)
                self.head_tape.forward_out(self._head_z, S.pos[s_ + 1], addend=S.pos[s_])
        else:
            # The autoregressive decode itself runs on the fused decoder kernels (positions only, nothing kept); with the
            # tokens known, every activation the backward needs is recomputed for all ph steps at once: causal
            # self-attention over the steps equals the KV-cached recurrence, and ph*B-row GEMMs replace ph B-row ones.
            if self._stash_decode_ok():
                self._ar_decode(stash=True)       # the pipeline decode keeps the stash itself: no second pass
            else:
                self._ar_decode()
                self._decoder_steps(0, ph)
        # loss value and dL/dpos in one pass (mgcvae.py:614-620)
        self.loss.zero_()
        K.l2_loss_bwd(S.pos[1:], self.y_st, S.G[1:], self.loss)
        return self.loss

    def _decoder_steps(self, s0, s1):
        """Decoder stack + head hidden layers for steps s0..s1-1 on the tokens in ``S.tok`` (step-major rows s*B+b), keeping
        every activation the backward reads; the keys / values of steps < s0 must already be in the stash.  (0, ph): all steps
        at once (tokens known); (s, s+1): one step of the recurrence.  Needs the encoder stash and ``S.mkv``."""
        S, B, T, d, H, ph = self.s, self.B, self.T, self.d, self.H, self.ph
        dec = self.dec
        layers = dec.transformer_decoder.layers
        R, r0, r1 = ph * B, s0 * B, s1 * B
        f2 = lambda t: t.view(R, t.shape[-1])[r0:r1]
        bm = lambda t: t[s0:s1].permute(1, 0, 2)          # (B, s1-s0, .) view of the step-major stash
        D = self.drop.get
        dr = lambda name: None if D(name) is None else (f2(D(name)[0]), D(name)[1])
        K.gemm(f2(S.tok), _t(dec.input_fc.weight), f2(S.x[0]), bias=dec.input_fc.bias, addend=S.pe_dec[r0:r1],
               drop=dr("dec.pe"), drop_after_add=True)
        mask_b = S.pad
        for l, lay in enumerate(layers):
            sa, ca = lay.self_attn, lay.multihead_attn
            K.gemm(f2(S.x[l]), _t(sa.in_proj_weight), f2(S.qkv[l]), bias=sa.in_proj_bias)
            ckv = S.qkv[l][0:s1].permute(1, 0, 2)          # keys / values of steps 0..s1-1
            pd1 = D(f"dec{l}.attn1")
            K.attention_fwd(bm(S.qkv[l])[:, :, 0:d], ckv[:, :, d:2 * d], ckv[:, :, 2 * d:], bm(S.a1[l]), S.p1[l][s0 * ph:], H,
                            self.scale, causal=True, causal_offset=s0, probs_strides=(ph * ph, ph),
                            pdrop=None if pd1 is None else (pd1[0][s0 * ph:], pd1[1]))
            tmp = S.tmp[r0:r1]
            K.gemm(f2(S.a1[l]), _t(sa.out_proj.weight), tmp, bias=sa.out_proj.bias, addend=f2(S.x[l]), drop=dr(f"dec{l}.drop1"))
            K.layernorm_fwd(tmp, None, lay.norm1.weight, lay.norm1.bias, f2(S.x1[l]), f2(S.xhat1[l]), S.rstd1[l].view(R)[r0:r1])
            K.gemm(f2(S.x1[l]), _t(ca.in_proj_weight[:d]), f2(S.cq[l]), bias=ca.in_proj_bias[:d])
            mkv = S.mkv[l].view(B, T, 2 * d)
            pd2 = D(f"dec{l}.attn2")
            K.attention_fwd(bm(S.cq[l]), mkv[:, :, :d], mkv[:, :, d:], bm(S.a2[l]), S.p2[l][s0 * T:], H, self.scale, pad=mask_b,
                            probs_strides=(ph * T, T), pdrop=None if pd2 is None else (pd2[0][s0 * T:], pd2[1]))
            K.gemm(f2(S.a2[l]), _t(ca.out_proj.weight), tmp, bias=ca.out_proj.bias, addend=f2(S.x1[l]), drop=dr(f"dec{l}.drop2"))
            K.layernorm_fwd(tmp, None, lay.norm2.weight, lay.norm2.bias, f2(S.x2[l]), f2(S.xhat2[l]), S.rstd2[l].view(R)[r0:r1])
            K.gemm(f2(S.x2[l]), _t(lay.linear1.weight), f2(S.f1[l]), bias=lay.linear1.bias, relu=True, drop=dr(f"dec{l}.ff"))
            K.gemm(f2(S.f1[l]), _t(lay.linear2.weight), tmp, bias=lay.linear2.bias, addend=f2(S.x2[l]), drop=dr(f"dec{l}.drop3"))
            K.layernorm_fwd(tmp, None, lay.norm3.weight, lay.norm3.bias, f2(S.x[l + 1]), f2(S.xhat3[l]),
                            S.rstd3[l].view(R)[r0:r1])
        no = self.n_out
        if self.map_mode:
            hcat = f2(S.hcat)
            K.gemm(f2(S.x[self.Ld]), _t(dec.output_fc.weight), hcat[:, :no], bias=dec.output_fc.bias)
            S.hcat[s0:s1, :, no:].copy_(S.enc_map_used.unsqueeze(0).expand(s1 - s0, B, -1))
            self.fuse_tape.forward_out(self.fuse_tape.forward_hidden(hcat, (r0, r1)), f2(S.u))
            self._head_z = self.head_tape.forward_hidden(f2(S.u), (r0, r1))
        else:
            K.gemm(f2(S.x[self.Ld]), _t(dec.output_fc.weight), f2(S.ho), bias=dec.output_fc.bias)
            self._head_z = self.head_tape.forward_hidden(f2(S.ho), (r0, r1))

    def _packed_decoder(self):
        if self._pdec is None:      # packed copies alias the parameters: built once, refreshed in place every step
            def dev_sd(m):
                return {k: v.detach() for k, v in m.state_dict(keep_vars=True).items()}
            self._pdec = ops.PackedDecoder(dev_sd(self.dec), self.H, dev_sd(self.head),
                                           dev_sd(self.fuse) if self.map_mode else None,
                                           dev_sd(self.memfuse) if self.lane_mode else None)
            self._pdec_fresh = True
        return self._pdec

    alias_memory_kv = True   # the memory's K/V stash = the decoder kernels' own prologue output (False: six GEMMs, tests)
    stash_decode = True      # let the pipeline decode keep the backward's stash (False: always the all-steps pass)

    def _stash_layout(self):
        """Layout of the decoder stash where the stash-keeping decode exists for this configuration (Basic / lane, autoregressive,
        the shipped widths); None otherwise."""
        if (self.teacher_forced or torch.device(self.device).type != "cuda" or len(self.head_tape_shape()) != 1
                or self.head_tape_shape()[0] != 32):
            return None
        if self.map_mode and hasattr(self.fuse.layer_seq, "lmlp_0"):       # (fusion/hist_map with hidden layers: no stash-keeping decode)
            return None
        lay = ops.decoder_stash_layout(self._packed_decoder(), self.B, self.T, self.ph, self._stash_map_dim())
        self.__dict__.pop("_pdec_fresh", None)       # (packed at construction: the first decode re-packs the then-current weights)
        return lay

    def _stash_map_dim(self):
        """width of the map feature that fusion/hist_map concatenates to output_fc's rows (0: no map)"""
        if not self.map_mode:
            return 0
        first = getattr(self.fuse.layer_seq, "lmlp_0", [self.fuse.layer_seq.out_mlp])[0]
        return int(first.weight.shape[1]) - self.n_out

    def head_tape_shape(self):
        """widths of decoder/MLP's hidden layers"""
        out, i = [], 0
        while hasattr(self.head.layer_seq, f"lmlp_{i}"):
            out.append(getattr(self.head.layer_seq, f"lmlp_{i}")[0].weight.shape[0])
            i += 1
        return out

    def _stash_decode_ok(self):
        """Per call (the device's state can change: a one-launch decode that timed out disables the pipeline)."""
        if not self.stash_decode or self.s.dec_stash is None:
            return False
        if self.drop.active and not self._ar_decode_takes_masks():
            return False
        return ops.decoder_stash_supported(self._packed_decoder(), self.B, self.T, self.ph, self.drop.active, self._stash_map_dim())

    # Measured in graph replay at B = 128: 3.44 ms per step with the packs forked to a second stream, 3.37 in line - a fork / join
    # inside a replayed HIP graph costs more (~35 us each way) than the two launches it hides (42 us).  Off.
    prepack = os.environ.get("MMT_PREPACK") is not None

    def _prepack(self):
        """The 4x4x1-order copies of the decoder's weights (forward pipeline) and their transposes (sweep pipeline) follow the
        parameters: re-laid out once per step.  Nothing before the decode reads them, so the two launches go to a second stream
        beside the history encoder's forward (a chain of small kernels that leaves most of the chip idle); ``_ar_decode`` joins."""
        self._packs_stream = None
        if not self.prepack or torch.device(self.device).type != "cuda" or self.teacher_forced or self._pdec is None:
            return
        cur = torch.cuda.current_stream(self.device)
        side = self.__dict__.get("_pack_stream")
        if side is None:
            side = self._pack_stream = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if not self.__dict__.pop("_pdec_fresh", False):
                self._pdec.repack()
            self._pdec_packed = True
            sp = self.__dict__.get("_sweep_pipe_state")
            if sp:
                sp[1].run()
                self._sweep_packed = True
        self._packs_stream = side

    def _join_packs(self):
        side = self.__dict__.get("_packs_stream")
        if side is not None:
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._packs_stream = None

    def _ar_decode_takes_masks(self):
        """Whether the inference decoder kernels can apply this step's keep masks (mmt_decoder_forward_dropout)."""
        ok = self.__dict__.get("_ar_masks_ok")
        if ok is None:
            ok = self.__dict__["_ar_masks_ok"] = (torch.device(self.device).type == "cuda" and
                                                  ops.decoder_dropout_supported(self._packed_decoder(), self.B, self.T, self.ph))
        return ok

    def _ar_decode(self, stash=False):
        """pos[1:] <- the 12-step autoregressive decode of the current parameters (mgcvae.py:375-449) on the fused
        decoder kernels, from the encoder output and start token already in the stash; with dropout active, on the
        launch-per-phase cluster kernels reading this step's keep masks."""
        S = self.s
        if self._pdec is None:      # packed copies alias the parameters: built once, refreshed in place every step
            def dev_sd(m):
                return {k: v.detach() for k, v in m.state_dict(keep_vars=True).items()}
            self._pdec = ops.PackedDecoder(dev_sd(self.dec), self.H, dev_sd(self.head),
                                           dev_sd(self.fuse) if self.map_mode else None,
                                           dev_sd(self.memfuse) if self.lane_mode else None)
        elif self.__dict__.pop("_pdec_packed", False):
            self._join_packs()      # (re-laid out beside the encoder's forward: _prepack)
        elif self.__dict__.pop("_pdec_fresh", False):
            pass                    # (built by _packed_decoder in this very step)
        else:
            self._pdec.repack()
        sites = None
        if self.drop.active:
            sites = {n: v for n, v in self.drop.view.items() if n.startswith("dec")}
        ops.decoder_forward(self._pdec, S.eh[self.Le].view(self.B, self.T, self.d), S.pad, start_pos=S.pos[0], ph=self.ph,
                            map_feat=S.enc_map_used if self.map_mode else None,
                            mem_feat=S.enc_lane[:, 0, :].contiguous() if self.lane_mode else None, pred_out=S.pred, dropout=sites,
                            stash=S.dec_stash if stash else None, workspace=self._dec_ws)
        S.pos[1:].copy_(S.pred.permute(1, 0, 2))

    # ------------------------------------------------------------------ backward
    def _wgrad(self, dY, X, w, bias=None):
        """gW = dY^T X over all rows, gb = column sums of dY."""
        self._wgrad_rows(dY, X, self.g(w), self.g(bias) if bias is not None else None)

    def _wgrad_rows(self, dY, X, gw, gb):
        R, M = dY.shape
        N = X.shape[1]
        if (R >= 512 and R % 4 == 0 and M % 4 == 0 and N % 4 == 0 and M >= 32 and N >= 32 and dY.stride(1) == 1
                and X.stride(1) == 1):      # (the split-K / group kernels walk K = the rows four at a time)
            # one pass: K range split over SPLITK workgroup slices, bias gradient from the same operand tiles
            sk = self.GROUP_SPLITK if self._deferred is not None else self.SPLITK
            kw = dict(A=dY.t(), B=X, Cm=gw, splitk=sk, part_stride=self.grad_total, asum=gb)
            if self._deferred is not None:
                self._deferred[0].append(kw)
            else:
                K.gemm(**kw)
            self._split_used = self._split_used or sk > 1
            return
        K.gemm(dY.t(), X, gw)
        if gb is not None:
            K.colsum(dY, gb)

    _deferred = None
    _split_used = False
    # K slices of a weight-gradient GEMM that goes out in a group launch: the slices exist to give a single launch enough
    # workgroups; the group launch has ~4000 tiles without them, each workgroup then walks its whole K range in one steady
    # pipeline, and there are no mirror buffers to sum afterwards (5.49 -> 5.38 ms per step at B = 128)
    GROUP_SPLITK = 1

    def _defer_wgrads(self):
        """From here to ``_flush_wgrads`` the split-K weight-gradient GEMMs and LayerNorm parameter reductions are collected
        instead of launched: they are independent of each other (each reads stashes that nothing rewrites before the flush and
        writes its own slice of the gradient buffer) and go out as ONE launch per kind -- ~70 GEMMs of 64..384 tiles each fill
        the chip together where each alone leaves most of it idle (``group_wgrads = False``: one launch each, same bits)."""
        if getattr(self, "group_wgrads", True):
            self._deferred = ([], [])

    def _flush_wgrads(self):
        if self._deferred is not None:
            gemms, lns = self._deferred
            self._deferred = None
            # the launch tables live in the engine's own cache: the Trainer's captured graphs hold their addresses, so they
            # must not be evicted while this engine (and with it those graphs) is alive
            K.gemm_group(gemms, cache=self._group_cache)
            K.ln_param_grads_group(lns, cache=self._group_cache)

    def _sum_split_parts(self, lo=0, hi=None):
        if self._split_used:
            t = self.grad_total
            hi = t if hi is None else hi
            if hi <= lo:
                return
            for z in range(1, self.SPLITK):
                K.add_inplace(self.flat_grad[lo:hi], self.flat_all[z * t + lo:z * t + hi])

    def _ln_grad(self, dy, xhat, ln):
        if dy.shape[0] >= 512:
            sk = self.GROUP_SPLITK if self._deferred is not None else self.SPLITK
            kw = dict(dy=dy, xhat=xhat, gw=self.g(ln.weight), gb=self.g(ln.bias), splitk=sk, part_stride=self.grad_total)
            if self._deferred is not None:
                self._deferred[1].append(kw)
            else:
                K.ln_param_grads(**kw)
            self._split_used = self._split_used or sk > 1
        else:
            K.ln_param_grads(dy, xhat, self.g(ln.weight), self.g(ln.bias))

    def _ln_gemm(self, dy, xhat, rstd, gamma, dx, w, out, ln_drop=None, **kw):
        """dx = LayerNorm backward of dy; out = epi(dx' @ w), dx' = dx or, with ``ln_drop=(keep, scale, dxm)``, dxm =
        keep*scale*dx.  One kernel when the row-block GEMM takes the shape (the per-step sweep at B <= 256), two otherwise."""
        M, Kd = dy.shape
        if M <= 256 and Kd % 16 == 0 and Kd <= 768:
            K.gemm(dy, w, out, ln_bwd=(xhat, rstd, gamma, dx), ln_drop=ln_drop, **kw)
        else:
            K.layernorm_bwd(dy, xhat, rstd, gamma, dx, drop=ln_drop)
            K.gemm(dx if ln_drop is None else ln_drop[2], w, out, **kw)

    def _ln_gemm_attn(self, dy, xhat, rstd, gamma, dx, w, d_o, q, k, v, probs, dq, dk, dv, pstrides, ln_drop=None, pdrop=None):
        """dx = LayerNorm backward of dy; d_o = dx' @ w (out_proj's input gradient, dx' as in ``_ln_gemm``); attention backward
        of one query per agent from d_o.  One kernel when the shapes allow (d_o then never leaves the chip), else the separate
        ones."""
        if K.gemm_attention_bwd_ok(dy, self.H, k.shape[1], self.dh):
            K.gemm_attention_bwd(dy, w, q, k, v, probs, dq, dk, dv, self.H, self.scale, ln_bwd=(xhat, rstd, gamma, dx),
                                 probs_strides=pstrides, ln_drop=ln_drop, pdrop=pdrop)
            return
        self._ln_gemm(dy, xhat, rstd, gamma, dx, w, d_o, ln_drop=ln_drop)
        K.attention_bwd(q, k, v, probs, d_o.view(dy.shape[0], 1, -1), dq, dk, dv, self.H, self.scale, probs_strides=pstrides,
                        pdrop=pdrop)

    def _site(self, name, s, dxm):
        """ln_drop triple of decoder site ``name`` at step ``s`` (None when the site is not dropped)."""
        v = self.drop.get(name, s)
        return None if v is None else (v[0], v[1], dxm)

    def _psite(self, name, off):
        v = self.drop.get(name)
        return None if v is None else (v[0][off:], v[1])

    def _sweep_chunks(self):
        """Agent ranges of the reverse-time sweep: multiples of 16 agents (the row-block GEMM's tile), at most ``sweep_ways``."""
        B = self.B
        ways = int(getattr(self, "sweep_ways", 1))
        if not torch.device(self.device).type == "cuda" or ways <= 1 or B < 64:
            return [(0, B)]
        n = max(1, min(ways, B // 32))
        per = ((B + n - 1) // n + 15) // 16 * 16
        return [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]

    def _sweep_streams(self, n):
        st = self.__dict__.setdefault("_sweep_st", [])
        while len(st) < n:
            st.append(torch.cuda.Stream(self.device))
        return st[:n]

    def _sweep(self, b0, b1):
        """Reverse-time sweep over the decoder steps for agents b0..b1-1."""
        S, B, T, d, H, ph = self.s, self.B, self.T, self.d, self.H, self.ph
        dec = self.dec
        layers = dec.transformer_decoder.layers
        L = self.Ld
        r = lambda t, s_: t[s_][b0:b1]                               # rows of step s_ of a step-major (ph, B, .) stash
        bq = lambda t, s_, c0, c1: t[s_:s_ + 1, b0:b1, c0:c1].permute(1, 0, 2)     # (agents, 1, .) query view
        bk = lambda t, s_, c0, c1: t[0:s_ + 1, b0:b1, c0:c1].permute(1, 0, 2)      # (agents, s_+1, .) key / value view
        site = lambda name, s_, dxm: None if self.drop.get(name) is None else \
            (self.drop.get(name)[0][s_][b0:b1], self.drop.get(name)[1], dxm)
        for s in range(ph - 1, -1, -1):
            # head: pos[s+1] = pos[s] + out(relu(LN(lin(.)))...)   (MLP.py:12-24, mgcvae.py:445)
            gin = self.head_tape.backward_step(s, r(S.G, s + 1), (b0, b1))
            if self.map_mode:
                self.fuse_tape.backward_step(s, gin, (b0, b1))          # -> gradient at cat[output_fc(x), encoded_map]
            K.gemm(r(S.gho, s), dec.output_fc.weight, r(S.gx[L], s))
            for l in range(L - 1, -1, -1):
                lay = layers[l]
                sa, ca = lay.self_attn, lay.multihead_attn
                # (each LayerNorm backward rides in the operand load of the GEMM that consumes it)
                dff = self.drop.get(f"dec{l}.ff")
                self._ln_gemm(r(S.gx[l + 1], s), r(S.xhat3[l], s), r(S.rstd3[l], s), lay.norm3.weight, r(S.gf2[l], s),
                              lay.linear2.weight, r(S.gf1[l], s), mask=r(S.f1[l], s), mask_scale=dff[1] if dff else 0.0,
                              ln_drop=site(f"dec{l}.drop3", s, r(S.gf2m[l], s)))
                K.gemm(r(S.gf1[l], s), lay.linear1.weight, r(S.gx2[l], s), addend=r(S.gf2[l], s))
                mkv, gmkv = S.mkv[l].view(B, T, 2 * d)[b0:b1], S.gmkv[l].view(B, T, 2 * d)[b0:b1]
                o2, o1 = b0 * H * ph * T + s * T, b0 * H * ph * ph + s * ph
                self._ln_gemm_attn(r(S.gx2[l], s), r(S.xhat2[l], s), r(S.rstd2[l], s), lay.norm2.weight, r(S.go2[l], s),
                                   ca.out_proj.weight, r(S.ga2[l], s),
                                   bq(S.cq[l], s, 0, d), mkv[:, :, :d], mkv[:, :, d:], S.p2[l][o2:],
                                   bq(S.gcq[l], s, 0, d), gmkv[:, :, :d], gmkv[:, :, d:], (ph * T, T),
                                   ln_drop=site(f"dec{l}.drop2", s, r(S.go2m[l], s)), pdrop=self._psite(f"dec{l}.attn2", o2))
                K.gemm(r(S.gcq[l], s), ca.in_proj_weight[:d], r(S.gx1[l], s), addend=r(S.go2[l], s))
                c, gc = S.qkv[l], S.gqkv[l]
                self._ln_gemm_attn(r(S.gx1[l], s), r(S.xhat1[l], s), r(S.rstd1[l], s), lay.norm1.weight, r(S.go1[l], s),
                                   sa.out_proj.weight, r(S.ga1[l], s),
                                   bq(c, s, 0, d), bk(c, s, d, 2 * d), bk(c, s, 2 * d, 3 * d), S.p1[l][o1:],
                                   bq(gc, s, 0, d), bk(gc, s, d, 2 * d), bk(gc, s, 2 * d, 3 * d), (ph * ph, ph),
                                   ln_drop=site(f"dec{l}.drop1", s, r(S.go1m[l], s)), pdrop=self._psite(f"dec{l}.attn1", o1))
                # (layer 0: x[0] = dropout(embedding + pe); the gradient that leaves the sweep is the one at the embedding)
                pe = self.drop.get("dec.pe") if l == 0 else None
                K.gemm(r(gc, s), sa.in_proj_weight, r(S.gx[l], s), addend=r(S.go1[l], s),
                       drop=None if pe is None else (pe[0][s][b0:b1], pe[1]), drop_after_add=True)
            if s > 0 and self.teacher_forced:      # G[s] = dL/dpos[s] + residual (cumsum) path; tokens are constants
                K.add_inplace(r(S.G, s), r(S.G, s + 1))
            elif s > 0:    # G[s] = dL/dpos[s] (already there) + residual path + embedding path
                K.gemm(r(S.gx[0], s), dec.input_fc.weight, r(S.G, s), addend=r(S.G, s + 1), accumulate=True)

    # Decoder weight gradients on a second stream beside the encoder's backward.  Measured inside the captured graph at B = 128:
    # 3.45 ms per step with, 3.43 without (the grouped launches already fill the chip; the chain's kernels only interleave): off.
    overlap_wgrads = False

    def backward(self):
        """Reverse-time sweep.  ``forward`` must have run; ``S.G[1:]`` holds dL/dpos from the loss kernel.
        Behind the sweep two things are independent of each other: the decoder's / head's deferred weight-gradient launches (a
        few large grouped launches) and the history encoder's backward (a chain of ~50 small dependent kernels that leaves most
        of the chip idle).  On a GPU they run side by side, the first on a second stream (a fork / join inside a captured graph)."""
        if not (self.overlap_wgrads and torch.device(self.device).type == "cuda"):
            self.backward_early()
            self.backward_late()
            return
        self._backward_sweep()
        cur = torch.cuda.current_stream(self.device)
        side = self.__dict__.get("_wgrad_stream")
        if side is None:
            side = self._wgrad_stream = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._backward_early_wgrads()
        self.backward_late()
        cur.wait_stream(side)

    # ------------------------------------------------------------------ the sweep as one pipeline launch (mmt_sweep_pipe_run)
    sweep_pipeline = True          # False: always the launch-per-kernel sweep (tests compare the two)

    def _sweep_pipe(self):
        """The descriptor of the one-launch sweep (csrc/decoder_bwd_pipe.cuh), or None where it does not apply: autoregressive
        graph (with or without dropout), Basic / Lane configuration, the shipped decoder geometry, decoder/MLP with one hidden
        layer."""
        sp = self.__dict__.get("_sweep_pipe_state")
        if sp is not None:
            return sp or None
        self._sweep_pipe_state = False
        dev = torch.device(self.device)
        self.sweep_pipe_why = "configuration (device / teacher forcing / map mode / dropout / head depth)"
        if dev.type != "cuda" or not self.sweep_pipeline or self.teacher_forced or len(self.head_tape.hidden) != 1:
            return None
        if self.map_mode and (self.fuse_tape.hidden or self.fuse_tape.k_in > 64):      # fusion/hist_map: a single Linear
            return None
        from . import _lib
        import ctypes as C
        S, d, H, L = self.s, self.d, self.H, self.Ld
        lin0, ln0 = self.head_tape.hidden[0]
        out = self.head_tape.out
        self.sweep_pipe_why = "head widths"
        if tuple(lin0.weight.shape) != (self.n_out, self.n_out) or tuple(out.weight.shape) != (self.n_pos, self.n_out):
            return None
        A = _lib.SweepPipe()
        A.B, A.T, A.ph, A.d, A.H, A.nlayers, A.d_ff, A.n_out, A.n_pos = self.B, self.T, self.ph, d, H, L, self.f_dec, self.n_out, self.n_pos
        A.gho_ld = self.n_out
        if self.map_mode:       # gradient at cat[output_fc(x), encoded_map] (fuse_tape.gin), at the fusion's output (head_tape.gin)
            A.w_fuse, A.h_gu = self.fuse_tape.out.weight.data_ptr(), self.head_tape.gin.data_ptr()
            A.gho_ld, A.map_dim = self.fuse_tape.k_in, self.fuse_tape.k_in - self.n_out
        lib = _lib.load()
        self.sweep_pipe_why = "geometry (mmt_sweep_pipe_workspace_floats)"
        if not lib.mmt_sweep_pipe_workspace_floats(C.byref(A)):
            return None
        DH, FS = d // H, self.f_dec // H
        pk = ops._Packer(dev)
        dec = self.dec
        holders = []
        for l, lay in enumerate(dec.transformer_decoder.layers):
            sa, ca = lay.self_attn, lay.multihead_attn
            wq, wk, wv = sa.in_proj_weight[:d], sa.in_proj_weight[d:2 * d], sa.in_proj_weight[2 * d:]
            holders.append(dict(
                w1=[pk.lin4_first(lay.linear2.weight, 1, 0, FS, H, tr=True), pk.lin4_first(ca.out_proj.weight, 1, 0, DH, H, tr=True),
                    pk.lin4_first(sa.out_proj.weight, 1, 0, DH, H, tr=True)],
                w2=[pk.lin4_second(lay.linear1.weight, FS, H, tr=True), pk.lin4_second(ca.in_proj_weight[:d], DH, H, tr=True),
                    pk.lin4_second(wq, DH, H, tr=True)],
                w2k=pk.lin4_second(wk, DH, H, tr=True), w2v=pk.lin4_second(wv, DH, H, tr=True)))
        ofc = pk.lin4_second(dec.output_fc.weight, self.n_out, 1, tr=True)
        self.sweep_pipe_why = "weight packs"
        packs = [ofc] + [h for hd in holders for h in (hd["w1"] + hd["w2"] + [hd["w2k"], hd["w2v"]])]
        if any(h is None for h in packs):          # (a holder is a one-element list, filled by finish())
            return None
        pk.finish()
        P = lambda t: t.data_ptr()
        for l, lay in enumerate(dec.transformer_decoder.layers):
            Wl, hd = A.layer[l], holders[l]
            for k in range(3):
                Wl.w1[k], Wl.w2[k] = hd["w1"][k][0], hd["w2"][k][0]
            Wl.w2k, Wl.w2v = hd["w2k"][0], hd["w2v"][0]
            for k, (ln, xh, rs, gdy, gdx) in enumerate(((lay.norm3, S.xhat3[l], S.rstd3[l], S.gx[l + 1], S.gf2m[l]),
                                                       (lay.norm2, S.xhat2[l], S.rstd2[l], S.gx2[l], S.go2m[l]),
                                                       (lay.norm1, S.xhat1[l], S.rstd1[l], S.gx1[l], S.go1m[l]))):
                Wl.gamma[k], Wl.xhat[k], Wl.rstd[k], Wl.gdy[k], Wl.gdx[k] = P(ln.weight), P(xh), P(rs), P(gdy), P(gdx)
            Wl.f1, Wl.gf1 = P(S.f1[l]), P(S.gf1[l])
            Wl.cq, Wl.mkv, Wl.p2, Wl.gcq, Wl.gmkv = P(S.cq[l]), P(S.mkv[l]), P(S.p2[l]), P(S.gcq[l]), P(S.gmkv[l])
            Wl.qkv, Wl.p1, Wl.gqkv = P(S.qkv[l]), P(S.p1[l]), P(S.gqkv[l])
            # training-mode dropout: the step's keep bytes where the reference's modules drop (None: that site is not dropped)
            D = self.drop.get
            for k, site in enumerate((f"dec{l}.drop3", f"dec{l}.drop2", f"dec{l}.drop1")):
                if D(site) is not None:
                    Wl.k_ln[k], Wl.s_ln[k] = P(D(site)[0]), D(site)[1]
            if D(f"dec{l}.ff") is not None:
                Wl.k_ff, Wl.s_ff = P(D(f"dec{l}.ff")[0]), D(f"dec{l}.ff")[1]
            for k, site in enumerate((f"dec{l}.attn2", f"dec{l}.attn1")):
                if D(site) is not None:
                    Wl.k_att[k], Wl.s_att[k] = P(D(site)[0]), D(site)[1]
        ht = self.head_tape
        A.G, A.gx0, A.w_in = P(S.G), P(S.gx[0]), P(dec.input_fc.weight)
        A.w_hout, A.h_gamma, A.w_h0, A.w_ofc = P(out.weight), P(ln0.weight), P(lin0.weight), ofc[0]
        A.h_hu, A.h_xhat, A.h_rstd = P(ht.hu[0]), P(ht.hxhat[0]), P(ht.hrstd[0])
        A.h_ghum, A.h_ghz, A.h_gho, A.poison = P(ht.ghum[0]), P(ht.ghz[0]), P(self.fuse_tape.gin if self.map_mode else ht.gin), P(self.loss)
        if self.drop.get("dec.pe") is not None:
            A.k_pe, A.s_pe = P(self.drop.get("dec.pe")[0]), self.drop.get("dec.pe")[1]
        self.sweep_pipe_why = "mmt_sweep_pipe_supported (device fit / disabled)"
        if not lib.mmt_sweep_pipe_supported(C.byref(A)):
            return None
        ws = torch.empty(int(lib.mmt_sweep_pipe_workspace_floats(C.byref(A))), dtype=torch.float32, device=dev)
        self.sweep_pipe_why = "in use"
        self._sweep_pipe_state = (A, pk, ws)
        return self._sweep_pipe_state

    def _sweep_on_pipeline(self):
        """True when the sweep went out as the one pipeline launch."""
        sp = self._sweep_pipe()
        if sp is None:
            return False
        from . import _lib
        import ctypes as C
        A, pk, ws = sp
        if not _lib.load().mmt_sweep_pipe_supported(C.byref(A)):      # e.g. the one-launch forms were disabled after a time-out
            return False
        self._join_packs()
        if not self.__dict__.pop("_sweep_packed", False):
            pk.run()                               # the transposed weight packs follow the parameters (one launch; else: _prepack)
        _lib.check(_lib.load().mmt_sweep_pipe_run(C.byref(A), C.c_void_p(ws.data_ptr()), ops._stream()), "mmt_sweep_pipe_run")
        return True

    def backward_early(self):
        """The sweep and every gradient of ``early_range`` (decoder, head, map encoder, fusion/hist_map)."""
        self._backward_sweep()
        self._backward_early_wgrads()

    def _backward_sweep(self):
        S, B = self.s, self.B
        S.acc_pool.zero_()
        # The sweep is a chain of ~6 small dependent kernels per (step, layer): latency, not throughput.  Agents are independent
        # of each other in it, so it can be cut into agent ranges on parallel streams (``sweep_ways`` > 1) -- measured on MI355X
        # inside the captured graph: 7.02 ms (1 way) / 7.76 (2) / 8.28 (4) per step at B = 128: every range's chain is as long
        # as the whole batch's (the kernels already spread the agents over workgroups), so the default is one range.
        chunks = self._sweep_chunks()
        if self._sweep_on_pipeline():
            pass
        elif len(chunks) == 1:
            self._sweep(0, B)
        else:
            cur = torch.cuda.current_stream(self.device)
            for (b0, b1), st in zip(chunks, self._sweep_streams(len(chunks))):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    self._sweep(b0, b1)
            for st in self._sweep_streams(len(chunks)):
                cur.wait_stream(st)

    def _backward_early_wgrads(self):
        """Deferred decoder weight gradients: one GEMM per Linear over all ph*B rows (reads what the sweep left, writes
        ``early_range`` of the flat gradient buffer only)."""
        S, B, T, d, H, ph = self.s, self.B, self.T, self.d, self.H, self.ph
        enc, dec = self.enc, self.dec
        layers = dec.transformer_decoder.layers
        L = self.Ld
        R = ph * B
        f2 = lambda t: t.view(R, t.shape[-1])
        self._defer_wgrads()
        self._wgrad(f2(S.gx[0]), f2(S.tok), dec.input_fc.weight, dec.input_fc.bias)
        memory = S.memf if self.lane_mode else S.eh[self.Le]          # what the cross-attention K/V were projected from
        for l, lay in enumerate(layers):
            sa, ca = lay.self_attn, lay.multihead_attn
            self._wgrad(f2(S.gqkv[l]), f2(S.x[l]), sa.in_proj_weight, sa.in_proj_bias)
            self._wgrad(f2(S.go1m[l]), f2(S.a1[l]), sa.out_proj.weight, sa.out_proj.bias)
            self._ln_grad(f2(S.gx1[l]), f2(S.xhat1[l]), lay.norm1)
            gw, gb = self.g(ca.in_proj_weight), self.g(ca.in_proj_bias)
            self._wgrad_rows(f2(S.gcq[l]), f2(S.x1[l]), gw[:d], gb[:d])
            self._wgrad_rows(S.gmkv[l], memory, gw[d:], gb[d:])
            self._wgrad(f2(S.go2m[l]), f2(S.a2[l]), ca.out_proj.weight, ca.out_proj.bias)
            self._ln_grad(f2(S.gx2[l]), f2(S.xhat2[l]), lay.norm2)
            self._wgrad(f2(S.gf1[l]), f2(S.x2[l]), lay.linear1.weight, lay.linear1.bias)
            self._wgrad(f2(S.gf2m[l]), f2(S.f1[l]), lay.linear2.weight, lay.linear2.bias)
            self._ln_grad(f2(S.gx[l + 1]), f2(S.xhat3[l]), lay.norm3)
        self._wgrad(f2(S.gho), f2(S.x[L]), dec.output_fc.weight, dec.output_fc.bias)
        if self.map_mode:
            self.head_tape.wgrads(self, f2(S.u), f2(S.G[1:]))
            self.fuse_tape.wgrads(self, f2(S.hcat), f2(self.head_tape.gin))
            self._flush_wgrads()          # (the map encoder's tapes re-use their gradient buffers: their GEMMs go out one by one)
            # d encoded_map = sum over the steps of the concatenation's last columns, then back through the CNN
            K.colsum(self.fuse_tape.gin.view(ph, B * self.fuse_tape.k_in), S.gmap)
            gm = S.gmap.view(B, self.fuse_tape.k_in)[:, self.n_out:]
            dm = self.drop.view.get("map.feature")
            if dm is not None:
                S.gmap_c.copy_(gm)
                K.dropout_apply(S.gmap_c, dm[0], dm[1], S.gmap_c)
                gm = S.gmap_c
            self.cnn_tape.backward(self, gm)
        else:
            self.head_tape.wgrads(self, f2(S.ho), f2(S.G[1:]))
            self._flush_wgrads()
        self._sum_split_parts(*self.early_range)

    def backward_late(self):
        """The history encoder's (and the lane branch's) gradients: ``late_ranges``, and the guard slot."""
        S, B, T, d, H, ph = self.s, self.B, self.T, self.d, self.H, self.ph
        enc, dec = self.enc, self.dec
        layers = dec.transformer_decoder.layers
        # ---- encoder: d memory = sum over decoder layers of d(memory K/V) W_kv
        gmem = S.gmemf if self.lane_mode else S.geh[self.Le]
        self._defer_wgrads()              # (every buffer the collected jobs read is per layer and written once per step)
        for l, lay in enumerate(layers):
            K.gemm(S.gmkv[l], lay.multihead_attn.in_proj_weight[d:], gmem, accumulate=(l > 0))
        if self.lane_mode:
            # back through fusion/lane_hist: its input was cat[memory, lane-0 feature repeated over the history]
            gcat = self.memfuse_tape.backward_step(0, S.gmemf)                     # (B*T, d + lane features)
            self.memfuse_tape.wgrads(self, S.memcat, S.gmemf)
            S.geh[self.Le].copy_(gcat[:, :d])
            S.glane.zero_()
            torch.sum(gcat.view(B, T, -1)[:, :, d:], dim=1, out=S.glane[:, 0, :])  # lanes 1, 2 never reach the loss (:404)
            self.lane_tape.backward(self, S.glane.view(B * self.n_lane, -1))
        for l in range(self.Le - 1, -1, -1):
            lay = enc.transformer_encoder.layers[l]
            sa = lay.self_attn
            De = self.drop.get
            d2, d1, dff = De(f"enc{l}.drop2"), De(f"enc{l}.drop1"), De(f"enc{l}.ff")
            K.layernorm_bwd(S.geh[l + 1], S.exhat2[l], S.erstd2[l], lay.norm2.weight, S.gef2[l],
                            drop=(d2[0], d2[1], S.gef2m[l]) if d2 else None)
            K.gemm(S.gef2m[l], lay.linear2.weight, S.gef1[l], mask=S.ef1[l], mask_scale=dff[1] if dff else 0.0)
            K.gemm(S.gef1[l], lay.linear1.weight, S.geh1[l], addend=S.gef2[l])
            K.layernorm_bwd(S.geh1[l], S.exhat1[l], S.erstd1[l], lay.norm1.weight, S.geo1[l],
                            drop=(d1[0], d1[1], S.geo1m[l]) if d1 else None)
            K.gemm(S.geo1m[l], sa.out_proj.weight, S.geatt[l])
            qkv, gq = S.eqkv[l].view(B, T, 3 * d), S.geqkv[l].view(B, T, 3 * d)
            K.attention_bwd(qkv[:, :, 0:d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], S.eprobs[l], S.geatt[l].view(B, T, d),
                            gq[:, :, 0:d], gq[:, :, d:2 * d], gq[:, :, 2 * d:], H, self.scale, pdrop=De(f"enc{l}.attn"))
            K.gemm(S.geqkv[l], sa.in_proj_weight, S.geh[l], addend=S.geo1[l], drop=De("enc.pe") if l == 0 else None,
                   drop_after_add=True)
            self._wgrad(S.gef2m[l], S.ef1[l], lay.linear2.weight, lay.linear2.bias)
            self._ln_grad(S.geh[l + 1], S.exhat2[l], lay.norm2)
            self._wgrad(S.gef1[l], S.eh1[l], lay.linear1.weight, lay.linear1.bias)
            self._ln_grad(S.geh1[l], S.exhat1[l], lay.norm1)
            self._wgrad(S.geo1m[l], S.eatt[l], sa.out_proj.weight, sa.out_proj.bias)
            self._wgrad(S.geqkv[l], S.eh[l], sa.in_proj_weight, sa.in_proj_bias)
        self._wgrad(S.geh[0], S.x0, enc.input_fc.weight, enc.input_fc.bias)
        self._flush_wgrads()
        for lo, hi in self.late_ranges:
            self._sum_split_parts(lo, min(hi, self.grad_total))
        self.guard.copy_(self.loss)

    # ------------------------------------------------------------------ convenience
    def predictions(self):
        """Standardised positions (B, ph, 2) of the last forward."""
        return self.s.pos[1:].permute(1, 0, 2).contiguous()


class _TrainLossFn(torch.autograd.Function):
    """Lets the reference's loop (``loss.backward(); clip; optimizer.step()``, train.py:244-253) drive the HIP
    backward: the parameters are the function's inputs, their gradients are slices of the engine's flat buffer."""

    @staticmethod
    def forward(ctx, engine, *params):
        ctx.engine = engine
        engine.forward()
        return engine.loss[0].clone()

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.engine
        eng.backward()
        eng.flat_grad.mul_(gout)
        return (None,) + tuple(eng.g(p) for _, p in eng.named_params)


def disable_dropout(model):
    """Set every dropout probability of ``model`` (a MultimodalGenerativeCVAE) to 0 -- nn.Dropout.p, nn.MultiheadAttention.dropout
    and the map-feature probability in the hyperparameters -- the configuration training parity is defined on (SURVEY.md
    section 0.5).  Engines read the probabilities when they are built: call this before the first training step."""
    for mod in model.node_modules.values():
        for m in mod.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, torch.nn.MultiheadAttention):
                m.dropout = 0.0
    cp = model.hyperparams.get("map_encoder", {}).get("cnn_param")
    if cp is not None:
        cp["dropout"] = 0.0
    model.__dict__.pop("_train_engines", None)
    return model


def train_loss_with_graph(model, x_st, y_st, ph, teacher_forced=False, maps=None, lanes=None):
    """``MultimodalGenerativeCVAE.train_loss`` body (Basic / Map_encoding_CNN / Lane_encoding_CNN): a scalar with autograd.
    Dropout as the modules say (training mode, p > 0) at the time the engine of this shape is first built."""
    B, T, _ = x_st.shape
    key = (B, T, ph, bool(teacher_forced), bool(model._m("node_history_encoder").training))
    cache = model.__dict__.setdefault("_train_engines", {})
    eng = cache.get(key)
    if eng is None:
        eng = cache[key] = TrainEngine(model, B, T, ph, teacher_forced=teacher_forced)
    eng.set_batch(x_st, y_st, maps, lanes)
    return _TrainLossFn.apply(eng, *[p for _, p in eng.named_params])


class Trainer:
    """Fused training step: forward + backward (one HIP graph), gradient all-reduce over RCCL when
    ``torch.distributed`` is initialised, clip-by-value + AdamW + ExponentialLR (second graph).

    With more than one rank the backward is captured in two graphs and the all-reduce in two buckets: the decoder-side gradients
    (``engine.early_range``, ~60 % of the bytes) are reduced asynchronously while the history encoder's backward still runs, the
    rest behind it (``overlap=False``: one bucket after the whole backward, as in round 1).

    Optimizer semantics are train.py:218-256: AdamW(lr=hp.learning_rate, betas .9/.999, eps 1e-8, weight decay
    0.01), second parameter group 'map_encoder' at lr 8e-4, ``clip_grad_value_(hp.grad_clip)`` on the averaged
    gradient, ExponentialLR(hp.learning_decay_rate) stepped once per batch."""

    def __init__(self, model, B, T, ph, use_graph=True, process_group=None, distributed=True, always_reduce=False,
                 dropout=None, seed=0, overlap=True):
        """``dropout``: None = as the modules say (training mode and p > 0, like torch), False = the dropout-free graph.
        ``seed``: of the dropout masks; data-parallel ranks must pass different seeds (or masks repeat across shards)."""
        import torch.distributed as dist
        hp = model.hyperparams
        if dropout is None and distributed and dist.is_available() and dist.is_initialized():
            seed = int(seed) + 7919 * dist.get_rank(process_group)
        self.engine = TrainEngine(model, B, T, ph, dropout=dropout, seed=seed)
        self.hp = hp
        self.world = 1
        if distributed and dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(process_group)
        # always_reduce: issue the all-reduce even in a one-rank group (tools/rehearse_rccl.py: the RCCL call between the two graph
        # replays on a single GPU)
        self.always_reduce = bool(always_reduce) and dist.is_available() and dist.is_initialized()
        self.pg = process_group
        self.lr0 = float(hp["learning_rate"])
        self.gamma = float(hp["learning_decay_rate"]) if hp.get("learning_rate_style", "exp") == "exp" else 1.0
        self.clip = float(hp["grad_clip"]) if hp.get("grad_clip") is not None else 0.0
        self.betas, self.eps, self.wd = (0.9, 0.999), 1e-8, 0.01
        dev = self.engine.device
        self.state = torch.tensor([1.0, self.lr0], dtype=torch.float32).to(dev)
        entries, self.moments = [], []
        for name, p in self.engine.named_params:
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            self.moments.append((m, v))
            scale = (0.0008 / self.lr0) if "map_encoder" in name else 1.0
            entries.append((p.data, self.engine.g(p), m, v, scale))
        self.table, self.ntensors = K.adamw_table(entries, dev)
        self.use_graph = use_graph
        self._g_fb = self._g_late = self._g_opt = None
        self.reduces = self.world > 1 or self.always_reduce
        self.overlap = bool(overlap) and self.reduces            # two backward graphs, two buckets
        self.steps_done = 0
        self.skipped = torch.zeros(1, dtype=torch.float32).to(dev)     # optimizer steps skipped because the loss was not finite
        self._skipped_seen = 0
        self.check_every = 50                # steps between host looks at ``skipped`` (a device -> host copy)
        self.time_allreduce = False          # bench: HIP events around the gradient all-reduce (allreduce_ms())
        self._ar_events = []
        if self.world > 1:
            self._sync_replicas(dist)

    def _sync_replicas(self, dist):
        """Data-parallel replicas must start identical (torch DDP broadcasts at construction; so does this): parameters from rank 0
        (AdamW moments start at zero on every rank), then a checksum over all ranks as the proof."""
        flat = torch.cat([p.data.reshape(-1) for _, p in self.engine.named_params])
        dist.broadcast(flat, src=0, group=self.pg)
        off = 0
        for _, p in self.engine.named_params:
            n = p.numel()
            p.data.copy_(flat[off:off + n].view_as(p.data))
            off += n
        chk = flat.double().sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        if float(lo) != float(hi):
            raise RuntimeError("Trainer: parameters differ across data-parallel ranks after the broadcast")

    def allreduce_ms(self):
        """Mean time per step that the gradient all-reduce adds between the backward and the optimizer, over the steps taken with
        ``time_allreduce`` set (None: never ran).  With ``overlap`` this is what is NOT hidden behind the encoder's backward:
        (first bucket issued -> last bucket reduced) minus the duration of that backward graph."""
        if not self._ar_events:
            return None
        torch.cuda.synchronize()
        tot = 0.0
        for ev in self._ar_events:
            tot += ev[0].elapsed_time(ev[1])
            if len(ev) == 4:
                tot -= ev[2].elapsed_time(ev[3])
        return max(tot, 0.0) / len(self._ar_events)

    def close(self):
        """Ordered teardown, to be called before ``torch.distributed.destroy_process_group()``: wait for everything
        this trainer enqueued -- the graph replays on the current stream AND the collectives on the process group's own stream,
        which ``work.wait()`` only orders the current stream behind --, then drop what references device or communicator state in
        a defined order: the last asynchronous Work, the timing events, the captured graphs (their exec objects are destroyed
        here, while the context is quiet, not whenever the garbage collector gets to them).  After close() the trainer cannot
        step any more (use_graph trainers would re-capture; this one refuses)."""
        if self.__dict__.get("_closed"):
            return
        dev = self.engine.device
        if torch.device(dev).type == "cuda":
            torch.cuda.synchronize(dev)
        w = self.__dict__.pop("_last_work", None)
        if w is not None:
            w.wait()
            del w
        self._ar_events = []
        for name in ("_g_opt", "_g_late", "_g_fb"):
            g = getattr(self, name, None)
            setattr(self, name, None)
            if g is not None:
                g.reset()
                del g
        if torch.device(dev).type == "cuda":
            torch.cuda.synchronize(dev)
        self._closed = True

    def _fwd_bwd(self):
        self.engine.forward()
        if self.overlap:
            self.engine.backward_early()
        else:
            self.engine.backward()

    def _bwd_late(self):
        self.engine.backward_late()

    def _opt(self):
        K.adamw_step(self.table, self.ntensors, self.state, self.betas[0], self.betas[1], self.eps, self.wd, self.clip,
                     1.0 / self.world, guard=self.engine.guard)
        K.lr_step(self.state, self.gamma, guard=self.engine.guard, skipped=self.skipped)

    def skipped_steps(self, check=True):
        """Optimizer steps skipped so far because the (all-reduced) loss was not finite -- e.g. a one-launch decode that gave up
        on a shared GPU leaves NaN positions (mmt_decoder_status).  ``check``: on a new skip, query and clear the decoder's
        fault state (which also keeps the one-launch forms off from now on) and warn."""
        n = int(self.skipped.item())
        if check and n > self._skipped_seen:
            import warnings
            why = "the loss was not finite"
            try:
                ops.decoder_status()
            except MmtError as e:
                why = str(e)
            warnings.warn(f"Trainer: {n - self._skipped_seen} optimizer step(s) skipped, weights untouched: {why}", RuntimeWarning)
            if self.use_graph:
                self._g_fb = self._g_late = self._g_opt = None      # re-capture: the decoder form baked into the graph may be the one that failed
        self._skipped_seen = n
        return n

    def _capture(self):
        # warm-up on a side stream (allocations, library loading), then capture both halves
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._fwd_bwd()
            if self.overlap:
                self._bwd_late()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # capture_error_mode "thread_local": with a process group up, ProcessGroupNCCL's watchdog thread queries its events
        # whenever it likes; under the default "global" mode such a call from ANOTHER thread invalidates this thread's capture
        # (seen on MI355X once in a handful of runs: the next launch fails and the watchdog aborts the process).  This thread
        # itself only launches kernels while capturing.
        mode = dict(capture_error_mode="thread_local")
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb, **mode):
            self._fwd_bwd()
        if self.overlap:
            self._g_late = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_late, **mode):
                self._bwd_late()
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, **mode):
            self._opt()

    def step(self, x_st, y_st, maps=None, lanes=None):
        """One optimisation step on this rank's shard; returns THIS RANK's loss (its shard's mean, not the global mean) as a
        1-element device tensor."""
        if self.__dict__.get("_closed"):
            raise MmtError("Trainer.step after close()")
        self.engine.set_batch(x_st, y_st, maps, lanes)
        if self.use_graph:
            if self._g_fb is None:
                self._capture()
            self._g_fb.replay()
        else:
            self._fwd_bwd()
        if self.overlap:
            import torch.distributed as dist
            eng = self.engine
            timed = self.time_allreduce and eng.flat_grad.is_cuda
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
            lo, hi = eng.early_range
            if timed:
                ev[0].record()
            work = dist.all_reduce(eng.flat_grad[lo:hi], group=self.pg, async_op=True) if hi > lo else None
            if timed:
                ev[2].record()
            if self.use_graph:
                self._g_late.replay()                 # the encoder's backward, beside the first bucket's all-reduce
            else:
                self._bwd_late()
            if timed:
                ev[3].record()
            if work is not None:
                work.wait()                           # (orders the current stream behind the collective; the host does not block)
                self._last_work = work                # (close() waits for it on the host before the group may be destroyed)
            for lo, hi in eng.late_ranges:
                dist.all_reduce(eng.flat_grad[lo:hi], group=self.pg)
            if timed:
                ev[1].record()
                self._ar_events.append(tuple(ev))
        elif self.reduces:
            import torch.distributed as dist
            if self.time_allreduce and self.engine.flat_grad.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                dist.all_reduce(self.engine.flat_grad, group=self.pg)
                ev[1].record()
                self._ar_events.append(ev)
            else:
                dist.all_reduce(self.engine.flat_grad, group=self.pg)      # one flat fp32 bucket
        if self.use_graph:
            self._g_opt.replay()
        else:
            self._opt()
        self.steps_done += 1
        if self.check_every and self.steps_done % self.check_every == 0:
            self.skipped_steps()
        # the kernels wrote the parameters through raw pointers (no torch version bump): drop the packed
        # inference copies so that the next predict() re-packs the updated weights
        for mod in (self.engine.enc, self.engine.dec, self.engine.head, self.engine.cnn, self.engine.fuse, self.engine.lane_enc,
                    self.engine.memfuse):
            if mod is None:
                continue
            mod.__dict__.pop("_pack_cache", None)
            mod.__dict__.pop("_pack_cache_dec", None)
        return self.engine.loss
