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

Dropout: the HIP training step implements the dropout-free graph, the configuration training parity is defined
on (the reference's dropout RNG stream is not a meaningful parity target).
"""
from __future__ import annotations

import math

import torch

from . import _train_kernels as K
from . import ops
from ._lib import MmtError


def _t(w):
    """nn.Linear weight (N,K) -> B operand (K,N) of ``X @ W^T``."""
    return w.t()


class _Stash:
    pass


class TrainEngine:
    """Forward-with-stash + backward for one (B, T, ph) shape of the Basic configuration."""

    def __init__(self, model, B, T, ph, device=None, teacher_forced=None):
        hp = model.hyperparams
        # non_at_dec (mgcvae.py:310-373): tokens come from the labels, positions are the running sum of the head
        # outputs and are NOT fed back, so no gradient flows through the tokens
        self.teacher_forced = (not hp.get("autoregressive", True)) if teacher_forced is None else bool(teacher_forced)
        if hp["map_cnn_encoding"] or hp["map_vit_encoding"] or hp["lane_cnn_encoding"]:
            raise MmtError("the HIP training step covers the Basic configuration (BASELINE config 2); "
                           "map / lane context encoders have no backward kernels yet")
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
        self.head_hidden = []
        i = 0
        while hasattr(self.head.layer_seq, f"lmlp_{i}"):
            seq = getattr(self.head.layer_seq, f"lmlp_{i}")
            self.head_hidden.append((seq[0], seq[1]))
            i += 1
        self.head_out = self.head.layer_seq.out_mlp
        if not isinstance(self.head_out, torch.nn.Linear):
            raise MmtError("decoder head must be a regression-mode Mlp")
        self._pdec = None
        self._collect_params(model)
        self._alloc()

    # ------------------------------------------------------------------ parameters / gradients
    def _collect_params(self, model):
        """(registrar key + parameter name, parameter) for every parameter that receives a gradient in the Basic
        graph; Encoder.output_fc and Trajectory_Decoder.mix_fc never do and are skipped by the
        optimizer exactly like torch skips ``grad is None``."""
        nt = model.node_type
        named = []
        for key, mod in ((nt + "/node_history_encoder", self.enc), (nt + "/decoder/transformer_decoder", self.dec),
                         (nt + "/decoder/MLP", self.head)):
            for pn, p in mod.named_parameters():
                if key.endswith("node_history_encoder") and pn.startswith("output_fc."):
                    continue
                if pn.startswith("mix_fc."):
                    continue
                named.append((f"{key}.{pn}", p))
        self.named_params = named
        total = 0
        offs = []
        for _, p in named:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise MmtError("parameters must be contiguous fp32")
            offs.append(total)
            total += (p.numel() + 63) & ~63
        # [0] is the gradient bucket (what the all-reduce and the optimizer see); [1..SPLITK-1] receive the other K-slices of
        # the split weight-gradient GEMMs at the same offsets and are summed into [0] at the end of backward()
        self.SPLITK = 4
        self.grad_total = total
        self.flat_all = torch.zeros(self.SPLITK * total, dtype=torch.float32, device=self.device)
        self.flat_grad = self.flat_all[:total]
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
        S.x = [b(ph, B, d) for _ in range(L + 1)]
        S.qkv = [b(ph, B, 3 * d) for _ in range(L)]
        S.p1 = [b(B * H * ph * ph) for _ in range(L)]      # (B,H,ph,ph) causal self-attention probabilities
        S.p2 = [b(B * H * ph * T) for _ in range(L)]       # (B,H,ph,T)
        for n in ("a1", "x1", "cq", "a2", "x2", "xhat1", "xhat2", "xhat3"):
            setattr(S, n, [b(ph, B, d) for _ in range(L)])
        for n in ("rstd1", "rstd2", "rstd3"):
            setattr(S, n, [b(ph, B) for _ in range(L)])
        S.f1 = [b(ph, B, fd) for _ in range(L)]
        S.mkv = [b(Re, 2 * d) for _ in range(L)]
        S.tmp = b(ph * B, d)
        no, nh = self.n_out, len(self.head_hidden)
        S.ho = b(ph, B, no)
        S.hu = [b(ph, B, no) for _ in range(nh)]
        S.hxhat = [b(ph, B, no) for _ in range(nh)]
        S.hrstd = [b(ph, B) for _ in range(nh)]
        S.htmp = b(ph * B, no)
        S.delta = b(ph, B, self.n_pos)
        S.pred = b(B, ph, self.n_pos)                      # autoregressive decode, agent-major (kernel output)
        # decoder gradients, step-major (kept for the deferred weight-gradient GEMMs)
        S.G = b(ph + 1, B, self.n_pos)
        S.gx = [b(ph, B, d) for _ in range(L + 1)]
        for n in ("gf2", "gx2", "go2", "ga2", "gcq", "gx1", "go1", "ga1"):
            setattr(S, n, [b(ph, B, d) for _ in range(L)])
        S.gf1 = [b(ph, B, fd) for _ in range(L)]
        S.gqkv, S.gmkv = [], []
        for _ in range(L):
            S.gqkv.append(S.acc_pool[o:o + n_gqkv].view(ph, B, 3 * d))
            o += n_gqkv
            S.gmkv.append(S.acc_pool[o:o + n_gmkv].view(Re, 2 * d))
            o += n_gmkv
        S.gho = b(ph, B, no)
        S.ghu = [b(ph, B, no) for _ in range(nh)]       # grad wrt relu(LN(.)) output
        S.ghum = [b(ph, B, no) for _ in range(nh)]      # the same, zeroed where relu was inactive
        S.ghz = [b(ph, B, no) for _ in range(nh)]       # grad wrt the Linear output feeding the LayerNorm
        self.s = S

    # ------------------------------------------------------------------ forward
    def set_batch(self, x_st, y_st):
        if tuple(x_st.shape) != tuple(self.x_st.shape) or tuple(y_st.shape) != tuple(self.y_st.shape):
            raise MmtError(f"batch shape {tuple(x_st.shape)} / {tuple(y_st.shape)} does not match the engine's "
                           f"{tuple(self.x_st.shape)} / {tuple(self.y_st.shape)}")
        self.x_st.copy_(x_st)
        self.y_st.copy_(y_st)

    def forward(self):
        """Encoder -> memory K/V -> ph decoder steps with position feedback; everything the backward needs is kept."""
        S, B, T, d, H, ph = self.s, self.B, self.T, self.d, self.H, self.ph
        enc, dec = self.enc, self.dec
        K.history_prepare(self.x_st, S.x0, S.pad)
        K.gemm(S.x0, _t(enc.input_fc.weight), S.eh[0], bias=enc.input_fc.bias, addend=S.pe_enc)
        for l, lay in enumerate(enc.transformer_encoder.layers):
            sa = lay.self_attn
            K.gemm(S.eh[l], _t(sa.in_proj_weight), S.eqkv[l], bias=sa.in_proj_bias)
            qkv = S.eqkv[l].view(B, T, 3 * d)
            K.attention_fwd(qkv[:, :, 0:d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], S.eatt[l].view(B, T, d), S.eprobs[l],
                            H, self.scale, pad=S.pad, causal=True)
            K.gemm(S.eatt[l], _t(sa.out_proj.weight), S.etmp, bias=sa.out_proj.bias, addend=S.eh[l])
            K.layernorm_fwd(S.etmp, None, lay.norm1.weight, lay.norm1.bias, S.eh1[l], S.exhat1[l], S.erstd1[l])
            K.gemm(S.eh1[l], _t(lay.linear1.weight), S.ef1[l], bias=lay.linear1.bias, relu=True)
            K.gemm(S.ef1[l], _t(lay.linear2.weight), S.etmp, bias=lay.linear2.bias, addend=S.eh1[l])
            K.layernorm_fwd(S.etmp, None, lay.norm2.weight, lay.norm2.bias, S.eh[l + 1], S.exhat2[l], S.erstd2[l])
        memory = S.eh[self.Le]
        layers = dec.transformer_decoder.layers
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
        if not self.teacher_forced:
            # The autoregressive decode itself runs on the fused decoder kernels (positions only, nothing kept); with the
            # tokens known, every activation the backward needs is recomputed for all ph steps at once: causal
            # self-attention over the steps equals the KV-cached recurrence, and ph*B-row GEMMs replace ph B-row ones.
            self._ar_decode()
        self._decoder_all_steps()
        if self.teacher_forced:     # positions = running sum of the head outputs (mgcvae.py:352,370), from pos[0] = 0
            R = ph * B
            z = (S.hu[-1] if self.head_hidden else S.ho).view(R, self.n_out)
            K.gemm(z, _t(self.head_out.weight), S.delta.view(R, self.n_pos), bias=self.head_out.bias)
            K.cumsum_axis(S.delta.view(1, ph, B * self.n_pos), S.pos[1:].view(1, ph, B * self.n_pos), 1)
        # loss value and dL/dpos in one pass (mgcvae.py:614-620)
        self.loss.zero_()
        K.l2_loss_bwd(S.pos[1:], self.y_st, S.G[1:], self.loss)
        return self.loss

    def _decoder_all_steps(self):
        """Decoder stack + head hidden layers for all ph steps at once on the tokens in ``S.tok`` (step-major rows
        s*B+b), keeping every activation the backward reads.  Needs the encoder stash and ``S.mkv``."""
        S, B, T, d, H, ph = self.s, self.B, self.T, self.d, self.H, self.ph
        dec = self.dec
        layers = dec.transformer_decoder.layers
        R = ph * B
        f2 = lambda t: t.view(R, t.shape[-1])
        K.gemm(f2(S.tok), _t(dec.input_fc.weight), f2(S.x[0]), bias=dec.input_fc.bias, addend=S.pe_dec)
        mask_b = S.pad
        for l, lay in enumerate(layers):
            sa, ca = lay.self_attn, lay.multihead_attn
            K.gemm(f2(S.x[l]), _t(sa.in_proj_weight), f2(S.qkv[l]), bias=sa.in_proj_bias)
            c = S.qkv[l].permute(1, 0, 2)                  # (B,ph,3d) view of the step-major stash
            K.attention_fwd(c[:, :, 0:d], c[:, :, d:2 * d], c[:, :, 2 * d:], S.a1[l].permute(1, 0, 2), S.p1[l], H,
                            self.scale, causal=True)
            K.gemm(f2(S.a1[l]), _t(sa.out_proj.weight), S.tmp, bias=sa.out_proj.bias, addend=f2(S.x[l]))
            K.layernorm_fwd(S.tmp, None, lay.norm1.weight, lay.norm1.bias, f2(S.x1[l]), f2(S.xhat1[l]), S.rstd1[l].view(R))
            K.gemm(f2(S.x1[l]), _t(ca.in_proj_weight[:d]), f2(S.cq[l]), bias=ca.in_proj_bias[:d])
            mkv = S.mkv[l].view(B, T, 2 * d)
            K.attention_fwd(S.cq[l].permute(1, 0, 2), mkv[:, :, :d], mkv[:, :, d:], S.a2[l].permute(1, 0, 2), S.p2[l], H,
                            self.scale, pad=mask_b)
            K.gemm(f2(S.a2[l]), _t(ca.out_proj.weight), S.tmp, bias=ca.out_proj.bias, addend=f2(S.x1[l]))
            K.layernorm_fwd(S.tmp, None, lay.norm2.weight, lay.norm2.bias, f2(S.x2[l]), f2(S.xhat2[l]), S.rstd2[l].view(R))
            K.gemm(f2(S.x2[l]), _t(lay.linear1.weight), f2(S.f1[l]), bias=lay.linear1.bias, relu=True)
            K.gemm(f2(S.f1[l]), _t(lay.linear2.weight), S.tmp, bias=lay.linear2.bias, addend=f2(S.x2[l]))
            K.layernorm_fwd(S.tmp, None, lay.norm3.weight, lay.norm3.bias, f2(S.x[l + 1]), f2(S.xhat3[l]),
                            S.rstd3[l].view(R))
        K.gemm(f2(S.x[self.Ld]), _t(dec.output_fc.weight), f2(S.ho), bias=dec.output_fc.bias)
        z = f2(S.ho)
        for i, (lin, ln) in enumerate(self.head_hidden):
            K.gemm(z, _t(lin.weight), S.htmp, bias=lin.bias)
            K.layernorm_fwd(S.htmp, None, ln.weight, ln.bias, f2(S.hu[i]), f2(S.hxhat[i]), S.hrstd[i].view(R), relu=True)
            z = f2(S.hu[i])

    def _ar_decode(self):
        """pos[1:] <- the 12-step autoregressive decode of the current parameters (mgcvae.py:375-449) on the fused
        decoder kernels, from the encoder output and start token already in the stash."""
        S = self.s
        if self._pdec is None:      # packed copies alias the parameters: built once, refreshed in place every step
            def dev_sd(m):
                return {k: v.detach() for k, v in m.state_dict(keep_vars=True).items()}
            self._pdec = ops.PackedDecoder(dev_sd(self.dec), self.H, dev_sd(self.head))
        else:
            self._pdec.repack()
        ops.decoder_forward(self._pdec, S.eh[self.Le].view(self.B, self.T, self.d), S.pad, start_pos=S.pos[0], ph=self.ph,
                            pred_out=S.pred)
        S.pos[1:].copy_(S.pred.permute(1, 0, 2))

    # ------------------------------------------------------------------ backward
    def _wgrad(self, dY, X, w, bias=None):
        """gW = dY^T X over all rows, gb = column sums of dY."""
        self._wgrad_rows(dY, X, self.g(w), self.g(bias) if bias is not None else None)

    def _wgrad_rows(self, dY, X, gw, gb):
        R, M = dY.shape
        N = X.shape[1]
        if R >= 512 and M % 4 == 0 and N % 4 == 0 and M >= 32 and N >= 32 and dY.stride(1) == 1 and X.stride(1) == 1:
            # one pass: K range split over SPLITK workgroup slices, bias gradient from the same operand tiles
            K.gemm(dY.t(), X, gw, splitk=self.SPLITK, part_stride=self.grad_total, asum=gb)
            self._split_used = True
            return
        K.gemm(dY.t(), X, gw)
        if gb is not None:
            K.colsum(dY, gb)

    def _sum_split_parts(self):
        if getattr(self, "_split_used", False):
            t = self.grad_total
            for z in range(1, self.SPLITK):
                K.add_inplace(self.flat_grad, self.flat_all[z * t:(z + 1) * t])

    def _ln_grad(self, dy, xhat, ln):
        if dy.shape[0] >= 512:
            K.ln_param_grads(dy, xhat, self.g(ln.weight), self.g(ln.bias), splitk=self.SPLITK, part_stride=self.grad_total)
            self._split_used = True
        else:
            K.ln_param_grads(dy, xhat, self.g(ln.weight), self.g(ln.bias))

    def _ln_gemm(self, dy, xhat, rstd, gamma, dx, w, out, **kw):
        """dx = LayerNorm backward of dy; out = epi(dx @ w).  One kernel when the row-block GEMM takes the shape (the
        per-step sweep at B <= 256), two otherwise."""
        M, Kd = dy.shape
        if M <= 256 and Kd % 16 == 0 and Kd <= 768:
            K.gemm(dy, w, out, ln_bwd=(xhat, rstd, gamma, dx), **kw)
        else:
            K.layernorm_bwd(dy, xhat, rstd, gamma, dx)
            K.gemm(dx, w, out, **kw)

    def _ln_gemm_attn(self, dy, xhat, rstd, gamma, dx, w, d_o, q, k, v, probs, dq, dk, dv, pstrides):
        """dx = LayerNorm backward of dy; d_o = dx @ w (out_proj's input gradient); attention backward of one query per
        agent from d_o.  One kernel when the shapes allow (d_o then never leaves the chip), else the separate ones."""
        if K.gemm_attention_bwd_ok(dy, self.H, k.shape[1], self.dh):
            K.gemm_attention_bwd(dy, w, q, k, v, probs, dq, dk, dv, self.H, self.scale, ln_bwd=(xhat, rstd, gamma, dx),
                                 probs_strides=pstrides)
            return
        self._ln_gemm(dy, xhat, rstd, gamma, dx, w, d_o)
        K.attention_bwd(q, k, v, probs, d_o.view(dy.shape[0], 1, -1), dq, dk, dv, self.H, self.scale, probs_strides=pstrides)

    def backward(self):
        """Reverse-time sweep.  ``forward`` must have run; ``S.G[1:]`` holds dL/dpos from the loss kernel."""
        S, B, T, d, H, ph = self.s, self.B, self.T, self.d, self.H, self.ph
        enc, dec = self.enc, self.dec
        layers = dec.transformer_decoder.layers
        L, nh = self.Ld, len(self.head_hidden)
        S.acc_pool.zero_()
        for s in range(ph - 1, -1, -1):
            # head: pos[s+1] = pos[s] + out(relu(LN(lin(.)))...)   (MLP.py:12-24, mgcvae.py:445)
            gin = S.G[s + 1]
            w = self.head_out.weight
            for i in range(nh - 1, -1, -1):
                lin, ln = self.head_hidden[i]
                K.gemm(gin, w, S.ghu[i][s])
                K.layernorm_bwd(S.ghu[i][s], S.hxhat[i][s], S.hrstd[i][s], ln.weight, S.ghz[i][s], y_relu=S.hu[i][s],
                                dy_masked=S.ghum[i][s])
                gin, w = S.ghz[i][s], lin.weight
            K.gemm(gin, w, S.gho[s])
            K.gemm(S.gho[s], dec.output_fc.weight, S.gx[L][s])
            for l in range(L - 1, -1, -1):
                lay = layers[l]
                sa, ca = lay.self_attn, lay.multihead_attn
                # (each LayerNorm backward rides in the operand load of the GEMM that consumes it)
                self._ln_gemm(S.gx[l + 1][s], S.xhat3[l][s], S.rstd3[l][s], lay.norm3.weight, S.gf2[l][s],
                              lay.linear2.weight, S.gf1[l][s], mask=S.f1[l][s])
                K.gemm(S.gf1[l][s], lay.linear1.weight, S.gx2[l][s], addend=S.gf2[l][s])
                mkv, gmkv = S.mkv[l].view(B, T, 2 * d), S.gmkv[l].view(B, T, 2 * d)
                self._ln_gemm_attn(S.gx2[l][s], S.xhat2[l][s], S.rstd2[l][s], lay.norm2.weight, S.go2[l][s],
                                   ca.out_proj.weight, S.ga2[l][s],
                                   S.cq[l][s:s + 1].permute(1, 0, 2), mkv[:, :, :d], mkv[:, :, d:], S.p2[l][s * T:],
                                   S.gcq[l][s:s + 1].permute(1, 0, 2), gmkv[:, :, :d], gmkv[:, :, d:], (ph * T, T))
                K.gemm(S.gcq[l][s], ca.in_proj_weight[:d], S.gx1[l][s], addend=S.go2[l][s])
                c, gc = S.qkv[l], S.gqkv[l]
                self._ln_gemm_attn(S.gx1[l][s], S.xhat1[l][s], S.rstd1[l][s], lay.norm1.weight, S.go1[l][s],
                                   sa.out_proj.weight, S.ga1[l][s],
                                   c[s:s + 1, :, 0:d].permute(1, 0, 2), c[0:s + 1, :, d:2 * d].permute(1, 0, 2),
                                   c[0:s + 1, :, 2 * d:].permute(1, 0, 2), S.p1[l][s * ph:],
                                   gc[s:s + 1, :, 0:d].permute(1, 0, 2), gc[0:s + 1, :, d:2 * d].permute(1, 0, 2),
                                   gc[0:s + 1, :, 2 * d:].permute(1, 0, 2), (ph * ph, ph))
                K.gemm(gc[s], sa.in_proj_weight, S.gx[l][s], addend=S.go1[l][s])
            if s > 0 and self.teacher_forced:      # G[s] = dL/dpos[s] + residual (cumsum) path; tokens are constants
                K.add_inplace(S.G[s], S.G[s + 1])
            elif s > 0:    # G[s] = dL/dpos[s] (already there) + residual path + embedding path
                K.gemm(S.gx[0][s], dec.input_fc.weight, S.G[s], addend=S.G[s + 1], accumulate=True)
        # ---- deferred decoder weight gradients: one GEMM per Linear over all ph*B rows
        R = ph * B
        f2 = lambda t: t.view(R, t.shape[-1])
        self._wgrad(f2(S.gx[0]), f2(S.tok), dec.input_fc.weight, dec.input_fc.bias)
        memory = S.eh[self.Le]
        for l, lay in enumerate(layers):
            sa, ca = lay.self_attn, lay.multihead_attn
            self._wgrad(f2(S.gqkv[l]), f2(S.x[l]), sa.in_proj_weight, sa.in_proj_bias)
            self._wgrad(f2(S.go1[l]), f2(S.a1[l]), sa.out_proj.weight, sa.out_proj.bias)
            self._ln_grad(f2(S.gx1[l]), f2(S.xhat1[l]), lay.norm1)
            gw, gb = self.g(ca.in_proj_weight), self.g(ca.in_proj_bias)
            self._wgrad_rows(f2(S.gcq[l]), f2(S.x1[l]), gw[:d], gb[:d])
            self._wgrad_rows(S.gmkv[l], memory, gw[d:], gb[d:])
            self._wgrad(f2(S.go2[l]), f2(S.a2[l]), ca.out_proj.weight, ca.out_proj.bias)
            self._ln_grad(f2(S.gx2[l]), f2(S.xhat2[l]), lay.norm2)
            self._wgrad(f2(S.gf1[l]), f2(S.x2[l]), lay.linear1.weight, lay.linear1.bias)
            self._wgrad(f2(S.gf2[l]), f2(S.f1[l]), lay.linear2.weight, lay.linear2.bias)
            self._ln_grad(f2(S.gx[l + 1]), f2(S.xhat3[l]), lay.norm3)
        self._wgrad(f2(S.gho), f2(S.x[L]), dec.output_fc.weight, dec.output_fc.bias)
        z = S.ho
        for i, (lin, ln) in enumerate(self.head_hidden):
            self._wgrad(f2(S.ghz[i]), f2(z), lin.weight, lin.bias)
            self._ln_grad(f2(S.ghum[i]), f2(S.hxhat[i]), ln)
            z = S.hu[i]
        self._wgrad(f2(S.G[1:]), f2(z), self.head_out.weight, self.head_out.bias)
        # ---- encoder: d memory = sum over decoder layers of d(memory K/V) W_kv
        gmem = S.geh[self.Le]
        for l, lay in enumerate(layers):
            K.gemm(S.gmkv[l], lay.multihead_attn.in_proj_weight[d:], gmem, accumulate=(l > 0))
        for l in range(self.Le - 1, -1, -1):
            lay = enc.transformer_encoder.layers[l]
            sa = lay.self_attn
            K.layernorm_bwd(S.geh[l + 1], S.exhat2[l], S.erstd2[l], lay.norm2.weight, S.gef2[l])
            K.gemm(S.gef2[l], lay.linear2.weight, S.gef1[l], mask=S.ef1[l])
            K.gemm(S.gef1[l], lay.linear1.weight, S.geh1[l], addend=S.gef2[l])
            K.layernorm_bwd(S.geh1[l], S.exhat1[l], S.erstd1[l], lay.norm1.weight, S.geo1[l])
            K.gemm(S.geo1[l], sa.out_proj.weight, S.geatt[l])
            qkv, gq = S.eqkv[l].view(B, T, 3 * d), S.geqkv[l].view(B, T, 3 * d)
            K.attention_bwd(qkv[:, :, 0:d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], S.eprobs[l], S.geatt[l].view(B, T, d),
                            gq[:, :, 0:d], gq[:, :, d:2 * d], gq[:, :, 2 * d:], H, self.scale)
            K.gemm(S.geqkv[l], sa.in_proj_weight, S.geh[l], addend=S.geo1[l])
            self._wgrad(S.gef2[l], S.ef1[l], lay.linear2.weight, lay.linear2.bias)
            self._ln_grad(S.geh[l + 1], S.exhat2[l], lay.norm2)
            self._wgrad(S.gef1[l], S.eh1[l], lay.linear1.weight, lay.linear1.bias)
            self._ln_grad(S.geh1[l], S.exhat1[l], lay.norm1)
            self._wgrad(S.geo1[l], S.eatt[l], sa.out_proj.weight, sa.out_proj.bias)
            self._wgrad(S.geqkv[l], S.eh[l], sa.in_proj_weight, sa.in_proj_bias)
        self._wgrad(S.geh[0], S.x0, enc.input_fc.weight, enc.input_fc.bias)
        self._sum_split_parts()

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


def train_loss_with_graph(model, x_st, y_st, ph, teacher_forced=False):
    """``MultimodalGenerativeCVAE.train_loss`` body for the Basic configuration: a scalar that carries autograd."""
    B, T, _ = x_st.shape
    key = (B, T, ph, bool(teacher_forced))
    cache = model.__dict__.setdefault("_train_engines", {})
    eng = cache.get(key)
    if eng is None:
        eng = cache[key] = TrainEngine(model, B, T, ph, teacher_forced=teacher_forced)
    eng.set_batch(x_st, y_st)
    return _TrainLossFn.apply(eng, *[p for _, p in eng.named_params])


class Trainer:
    """Fused training step: forward + backward (one HIP graph), gradient all-reduce over RCCL when
    ``torch.distributed`` is initialised, clip-by-value + AdamW + ExponentialLR (second graph).

    Optimizer semantics are train.py:218-256: AdamW(lr=hp.learning_rate, betas .9/.999, eps 1e-8, weight decay
    0.01), second parameter group 'map_encoder' at lr 8e-4, ``clip_grad_value_(hp.grad_clip)`` on the averaged
    gradient, ExponentialLR(hp.learning_decay_rate) stepped once per batch."""

    def __init__(self, model, B, T, ph, use_graph=True, process_group=None, distributed=True):
        import torch.distributed as dist
        hp = model.hyperparams
        self.engine = TrainEngine(model, B, T, ph)
        self.hp = hp
        self.world = 1
        if distributed and dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(process_group)
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
        self._g_fb = self._g_opt = None
        self.steps_done = 0

    def _fwd_bwd(self):
        self.engine.forward()
        self.engine.backward()

    def _opt(self):
        K.adamw_step(self.table, self.ntensors, self.state, self.betas[0], self.betas[1], self.eps, self.wd, self.clip,
                     1.0 / self.world)
        K.lr_step(self.state, self.gamma)

    def _capture(self):
        # warm-up on a side stream (allocations, library loading), then capture both halves
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb):
            self._fwd_bwd()
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt):
            self._opt()

    def step(self, x_st, y_st):
        """One optimisation step on this rank's shard; returns the local loss as a 1-element device tensor."""
        self.engine.set_batch(x_st, y_st)
        if self.use_graph:
            if self._g_fb is None:
                self._capture()
            self._g_fb.replay()
        else:
            self._fwd_bwd()
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.engine.flat_grad, group=self.pg)      # one flat fp32 bucket
        if self.use_graph:
            self._g_opt.replay()
        else:
            self._opt()
        self.steps_done += 1
        # the kernels wrote the parameters through raw pointers (no torch version bump): drop the packed
        # inference copies so that the next predict() re-packs the updated weights
        for mod in (self.engine.enc, self.engine.dec, self.engine.head):
            mod.__dict__.pop("_pack_cache", None)
            mod.__dict__.pop("_pack_cache_dec", None)
        return self.engine.loss
