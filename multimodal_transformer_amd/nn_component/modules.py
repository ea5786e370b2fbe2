"""Parameter containers + HIP forwards for the hot-path modules.

State_dict keys must equal the reference's so that its checkpoints load
(SURVEY.md §8b): the stock ``torch.nn`` containers below (``nn.Linear``,
``nn.TransformerEncoder`` ...) are used *only* as named parameter holders with
the same default initialisation order as the reference -- same seed, same
initial weights -- and are never called.  The arithmetic of every ``forward``
here is one C-ABI call into libmmt_amd.so.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops, torch_ops
from .._lib import MmtError


class _PackedMixin:
    """Caches the packed-weight descriptor until a parameter is modified (version counters)."""

    def _packed(self, builder):
        # (parameters() / buffers() walk the same tensors as state_dict() at a third of its host cost: no dict, no key strings -
        # this check runs on every predict)
        key = ops._version_key(self.parameters()) + ops._version_key(self.buffers())
        cached = getattr(self, "_pack_cache", None)
        if cached is None or cached[0] != key:
            sd = {k: v.detach() for k, v in self.state_dict(keep_vars=True).items()}
            for k, v in sd.items():
                if not v.is_cuda:
                    raise MmtError(f"parameter {k} is on {v.device}: move the module to the GPU "
                                   "(the kernels have no CPU fallback)")
            cached = (key, builder({k: v.contiguous().float() for k, v in sd.items()}))
            object.__setattr__(self, "_pack_cache", cached)
        return cached[1]

    def __getstate__(self):          # never pickle device descriptors into checkpoints
        st = self.__dict__.copy()
        st.pop("_pack_cache", None)
        return st


class PositionalEncoding(nn.Module):
    """Sinusoid table buffer ``pe`` (Transformer_model.py:112-129); added inside the kernels."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(0, max_len).unsqueeze(1).float()
        freq = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * freq)
        pe[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", pe)

    def forward(self, x):
        """x (..., L, d) + pe[:L] (Transformer_model.py:126-129), dropout-free.  The encoder / decoder kernels add the table
        themselves; this stand-alone form is two strided copy launches (``mmt_copy3``)."""
        from .. import _train_kernels as K
        if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != torch.float32 or x.dim() < 2:
            raise MmtError("PositionalEncoding.forward needs a float32 device tensor (..., L, d) (no CPU fallback)")
        L, d = x.shape[-2], x.shape[-1]
        if L > self.pe.shape[0] or d != self.pe.shape[1] or x.numel() == 0:
            raise MmtError(f"PositionalEncoding.forward: {L} positions x {d} features against a {tuple(self.pe.shape)} table")
        src = x.contiguous().view(-1, L, d)
        y = torch.empty_like(src)
        K.copy3(src, y)
        K.copy3(self.pe[:L].unsqueeze(0).expand(src.shape[0], L, d), y, accumulate=True)
        return y.view(x.shape)


class Encoder(_PackedMixin, nn.Module):
    """History encoder (Transformer_model.py:11-59)."""

    def __init__(self, ninp=8, ntimestep=9, in_dim=128, nhead=2, fdim=256, nlayers=6, noutput=32, dropout=0.2,
                 low_dim="sum"):
        super().__init__()
        self.model_type = "Transformer"
        self.input_fc = nn.Linear(ninp, in_dim)
        self.pos_encoder = PositionalEncoding(in_dim, dropout)
        layer = nn.TransformerEncoderLayer(in_dim, nhead, fdim, dropout, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(layer, nlayers)      # parameter holder only
        self.low_dim = low_dim
        self.output_fc = nn.Linear(in_dim if low_dim == "sum" else in_dim * ntimestep, noutput)  # unused, in ckpt
        self.nhead = nhead

    def init_weights(self) -> None:
        """Transformer_model.py:38-43 (never called by the reference's constructors: the call is commented out there)."""
        torch.nn.init.zeros_(self.input_fc.bias)
        torch.nn.init.xavier_normal_(self.input_fc.weight)
        torch.nn.init.zeros_(self.output_fc.bias)
        torch.nn.init.xavier_normal_(self.output_fc.weight)

    def _nhead(self):
        return getattr(self, "nhead", None) or self.transformer_encoder.layers[0].self_attn.num_heads

    def packed(self):
        nh = self._nhead()
        return self._packed(lambda sd: ops.PackedEncoder(sd, nh))

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        """src (B,T,ninp) with NaN padding.  The causal mask and the ``isnan(src)[:,:,1]`` key padding mask
        are what mgcvae.py:276-282 always passes; they are rebuilt inside the kernel, so the two mask
        arguments are accepted for signature compatibility and ignored."""
        return torch_ops.encoder_fwd(self.packed(), src.contiguous())[0]

    def forward_with_mask(self, src):
        return torch_ops.encoder_fwd(self.packed(), src.contiguous())


class Decoder(_PackedMixin, nn.Module):
    """Base decoder parameter layout (Transformer_model.py:62-109)."""

    def __init__(self, nlayers, ninp, in_dim, nhead, fdim, noutput, dropout=0.2):
        super().__init__()
        self.model_type = "Transformer"
        self.input_fc = nn.Linear(ninp, in_dim)
        self.pos_encoder = PositionalEncoding(in_dim, dropout)
        layer = nn.TransformerDecoderLayer(in_dim, nhead, fdim, dropout, batch_first=True)
        self.transformer_decoder = nn.TransformerDecoder(layer, nlayers)      # parameter holder only
        self.output_fc = nn.Linear(in_dim, noutput)
        self.nhead = nhead

    def init_weights(self) -> None:
        """Transformer_model.py:83-89 (never called by the reference's constructors: the call is commented out there)."""
        torch.nn.init.zeros_(self.input_fc.bias)
        torch.nn.init.xavier_normal_(self.input_fc.weight)
        torch.nn.init.zeros_(self.output_fc.bias)
        torch.nn.init.xavier_normal_(self.output_fc.weight)


class Lane_Decoder(Decoder):
    """Transformer_model.py:132-152 (exported by the reference's nn_component, instantiated nowhere): a decoder pass WITHOUT a
    target mask - every lane token attends to all lane tokens, then to the history memory.  Not on the hot path, so it runs on
    the whole-batch kernels (``mmt_gemm`` / ``mmt_attention_fwd`` / ``mmt_layernorm_fwd``), dropout-free."""

    def __init__(self, ninp, in_dim, nhead, fdim, nlayers, noutput):
        super().__init__(nlayers, ninp, in_dim, nhead, fdim, noutput)

    def forward(self, tgt, memory, memory_key_padding_mask=None):
        from .. import _train_kernels as K
        for t, name in ((tgt, "tgt"), (memory, "memory")):
            if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3:
                raise MmtError(f"Lane_Decoder.forward: {name} must be a float32 device tensor (B, L, features) (no CPU fallback)")
        B, Lq, _ = tgt.shape
        T, d = memory.shape[1], self.input_fc.weight.shape[0]
        H = self.nhead
        if memory.shape[0] != B or memory.shape[2] != d or B == 0 or Lq == 0 or T == 0:
            raise MmtError(f"Lane_Decoder.forward: tgt {tuple(tgt.shape)} / memory {tuple(memory.shape)} do not line up with d_model {d}")
        dev = tgt.device
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        pad = None
        if memory_key_padding_mask is not None:
            pad = memory_key_padding_mask.to(torch.uint8).contiguous()
        scale = 1.0 / math.sqrt(d // H)
        M = B * Lq
        u = new(M, d)
        K.gemm(tgt.contiguous().view(M, -1), self.input_fc.weight.t(), u, bias=self.input_fc.bias)
        x = self.pos_encoder(u.view(B, Lq, d)).view(M, d)
        mem = memory.contiguous().view(B * T, d)

        def add_norm(sub, res, ln):
            y = new(M, d)
            K.layernorm_fwd(sub, res, ln.weight, ln.bias, y, new(M, d), new(M))
            return y

        for layer in self.transformer_decoder.layers:
            sa, ca = layer.self_attn, layer.multihead_attn
            qkv = new(M, 3 * d)
            K.gemm(x, sa.in_proj_weight.t(), qkv, bias=sa.in_proj_bias)
            q3 = qkv.view(B, Lq, 3 * d)
            o = new(B, Lq, d)
            K.attention_fwd(q3[:, :, 0:d], q3[:, :, d:2 * d], q3[:, :, 2 * d:], o, new(B, H, Lq, Lq), H, scale)
            sub = new(M, d)
            K.gemm(o.view(M, d), sa.out_proj.weight.t(), sub, bias=sa.out_proj.bias)
            x = add_norm(sub, x, layer.norm1)
            q = new(M, d)
            K.gemm(x, ca.in_proj_weight[:d].t(), q, bias=ca.in_proj_bias[:d].contiguous())
            kv = new(B * T, 2 * d)
            K.gemm(mem, ca.in_proj_weight[d:].t(), kv, bias=ca.in_proj_bias[d:].contiguous())
            kv3 = kv.view(B, T, 2 * d)
            o = new(B, Lq, d)
            K.attention_fwd(q.view(B, Lq, d), kv3[:, :, 0:d], kv3[:, :, d:], o, new(B, H, Lq, T), H, scale, pad=pad)
            sub = new(M, d)
            K.gemm(o.view(M, d), ca.out_proj.weight.t(), sub, bias=ca.out_proj.bias)
            x = add_norm(sub, x, layer.norm2)
            hid = new(M, layer.linear1.weight.shape[0])
            K.gemm(x, layer.linear1.weight.t(), hid, bias=layer.linear1.bias, relu=True)
            sub = new(M, d)
            K.gemm(hid, layer.linear2.weight.t(), sub, bias=layer.linear2.bias)
            x = add_norm(sub, x, layer.norm3)
        out = new(M, self.output_fc.weight.shape[0])
        K.gemm(x, self.output_fc.weight.t(), out, bias=self.output_fc.bias)
        return out.view(B, Lq, -1)


class Trajectory_Decoder(Decoder):
    """Trajectory decoder (Transformer_model.py:155-199)."""

    def __init__(self, tgt_inp, lane_inp, in_dim, nhead, fdim, nlayers, noutput):
        super().__init__(nlayers, tgt_inp, in_dim, nhead, fdim, noutput)
        self.mix_fc = nn.Linear(lane_inp + in_dim, in_dim)

    def _nhead(self):
        return getattr(self, "nhead", None) or self.transformer_decoder.layers[0].self_attn.num_heads

    def packed(self, head: "Mlp | None" = None, map_fusion: "Mlp | None" = None, mem_fusion: "Mlp | None" = None,
               use_mix_fc: bool = False):
        """Descriptor of this decoder together with the small Mlps that run inside the same kernel."""
        mods = [m for m in (head, map_fusion, mem_fusion) if m is not None]
        tensors = list(self.state_dict(keep_vars=True).values())
        for m in mods:
            tensors += list(m.state_dict(keep_vars=True).values())
        key = (ops._version_key(tensors), use_mix_fc, tuple(id(m) for m in (head, map_fusion, mem_fusion)))
        cache = self.__dict__.setdefault("_pack_cache_dec", {})
        hit = cache.get(key[1:])
        if hit is not None and hit[0] == key[0]:
            return hit[1]

        def dev_sd(m):
            sd = {k: v.detach().contiguous().float() for k, v in m.state_dict(keep_vars=True).items()}
            for k, v in sd.items():
                if not v.is_cuda:
                    raise MmtError(f"parameter {k} is on {v.device}: move the module to the GPU "
                                   "(the kernels have no CPU fallback)")
            return sd

        sd = dev_sd(self)
        memf = None
        if use_mix_fc:
            memf = {"weight": sd["mix_fc.weight"], "bias": sd["mix_fc.bias"]}
        elif mem_fusion is not None:
            memf = dev_sd(mem_fusion)
        packed = ops.PackedDecoder(sd, self._nhead(), dev_sd(head) if head is not None else None,
                                   dev_sd(map_fusion) if map_fusion is not None else None, memf,
                                   mem_fusion_is_mix_fc=use_mix_fc)
        cache[key[1:]] = (key[0], packed)
        return packed

    def __getstate__(self):
        st = super().__getstate__()
        st.pop("_pack_cache_dec", None)
        return st

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, memory_key_padding_mask=None, lane_feature=None):
        """Teacher-forced pass over ``tgt`` (B,s,2): causal self-attention, cross-attention masked by key
        padding only (``memory_mask`` is dropped exactly like the reference, :192-197).  ``tgt_mask`` is always
        the causal mask in the reference's call sites and is rebuilt in the kernel."""
        pad = None
        if memory_key_padding_mask is not None:
            pad = memory_key_padding_mask.to(torch.uint8).contiguous()
        mem_feat = None
        if lane_feature is not None:
            # the reference passes the lane feature already repeated over T (mgcvae.py:350-351): every row equal
            mem_feat = lane_feature[:, 0, :].contiguous()
        dec = self.packed(use_mix_fc=lane_feature is not None)
        h = torch_ops.decoder_tf_fwd(dec, memory.contiguous(), pad, tgt.contiguous(), mem_feat)
        return h


class Mlp(_PackedMixin, nn.Module):
    """(Linear, LayerNorm, ReLU) x (layer_num-1) + Linear [+ Softmax()]   (MLP.py:6-30)."""

    def __init__(self, in_channels, output_size, layer_num, mode):
        super().__init__()
        self.layer_seq = nn.Sequential()
        for i in range(layer_num - 1):
            self.layer_seq.add_module(
                f"lmlp_{i}", nn.Sequential(nn.Linear(in_channels, in_channels), nn.LayerNorm(in_channels), nn.ReLU()))
        if mode == "classification":
            self.layer_seq.add_module("out_mlp", nn.Sequential(nn.Linear(in_channels, output_size), nn.Softmax()))
        elif mode == "regression":
            self.layer_seq.add_module("out_mlp", nn.Linear(in_channels, output_size))
        self.mode = mode

    def forward(self, x):
        """Stand-alone call (MLP.py:28-30) on the whole-batch kernels: per hidden layer one GEMM (+bias) and one
        LayerNorm+ReLU launch, then the output GEMM; classification mode ends in ``nn.Softmax()`` with dim=None,
        which torch resolves to dim 0 for 0-, 1- and 3-D inputs and dim 1 otherwise (so the (B, ph, 96) tensor
        the reference feeds it is normalised over the BATCH).  Inside the decoders the same
        modules run fused (``Trajectory_Decoder.packed(head=..., map_fusion=..., mem_fusion=...)``)."""
        from .. import _train_kernels as K
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise MmtError("Mlp.forward needs a device tensor (no CPU fallback)")
        if x.dtype != torch.float32:
            raise MmtError(f"Mlp.forward: dtype {x.dtype}, expected torch.float32")
        shape = x.shape
        z = x.contiguous().view(-1, shape[-1])
        rows = z.shape[0]
        if rows == 0:
            raise MmtError("Mlp.forward: empty input")
        i = 0
        while hasattr(self.layer_seq, f"lmlp_{i}"):
            lin, ln = getattr(self.layer_seq, f"lmlp_{i}")[0], getattr(self.layer_seq, f"lmlp_{i}")[1]
            if z.shape[1] != lin.weight.shape[1]:
                raise MmtError(f"Mlp.forward: {z.shape[1]} input features, layer expects {lin.weight.shape[1]}")
            u = torch.empty(rows, lin.weight.shape[0], dtype=torch.float32, device=x.device)
            K.gemm(z, lin.weight.t(), u, bias=lin.bias)
            y = torch.empty_like(u)
            K.layernorm_fwd(u, None, ln.weight, ln.bias, y, torch.empty_like(u),
                            torch.empty(rows, dtype=torch.float32, device=x.device), relu=True)
            z = y
            i += 1
        out = self.layer_seq.out_mlp
        lin = out if isinstance(out, nn.Linear) else out[0]
        if z.shape[1] != lin.weight.shape[1]:
            raise MmtError(f"Mlp.forward: {z.shape[1]} input features, output layer expects {lin.weight.shape[1]}")
        y = torch.empty(rows, lin.weight.shape[0], dtype=torch.float32, device=x.device)
        K.gemm(z, lin.weight.t(), y, bias=lin.bias)
        y = y.view(*shape[:-1], lin.weight.shape[0])
        if not isinstance(out, nn.Linear):
            axis = 0 if y.dim() in (0, 1, 3) else 1           # torch.nn.functional._get_softmax_dim
            sm = torch.empty_like(y)
            K.softmax_axis(y, sm, axis)
            y = sm
        return y


class CNNMapEncoder(_PackedMixin, nn.Module):
    """Map CNN (map_encoder.py:6-36)."""

    def __init__(self, map_channels, hidden_channels, output_size, masks, strides, patch_size):
        super().__init__()
        self.convs = nn.ModuleList()
        self.in_hw = (patch_size[0] + patch_size[2], patch_size[1] + patch_size[3])
        h, w = self.in_hw
        for i, hc in enumerate(hidden_channels):
            self.convs.append(nn.Conv2d(map_channels if i == 0 else hidden_channels[i - 1], hc, masks[i],
                                        stride=strides[i]))
            h = (h - masks[i]) // strides[i] + 1
            w = (w - masks[i]) // strides[i] + 1
        self.fc = nn.Linear(hidden_channels[-1] * h * w, output_size)
        self.strides = list(strides)

    def _strides(self):
        return getattr(self, "strides", None) or [c.stride[0] for c in self.convs]

    def _in_hw(self, x):
        return getattr(self, "in_hw", None) or tuple(x.shape[-2:])

    def _pk(self, x):
        strides, in_hw = self._strides(), self._in_hw(x)
        return self._packed(lambda sd: ops.PackedCnn(sd, strides, in_hw))

    def forward(self, x, training=False):
        """x is the already rescaled ``map*2-1`` raster, as in the reference's call (mgcvae.py:289-290)."""
        return torch_ops.cnn_map_fwd(self._pk(x), x.contiguous(), False)

    def forward_raw(self, maps):
        """maps in [0,1]; the ``*2-1`` rescale is fused into the first conv's loads (saves one pass over HBM)."""
        return torch_ops.cnn_map_fwd(self._pk(maps), maps.contiguous(), True)


class Lane_Encoder(_PackedMixin, nn.Module):
    """Lane point encoder (Lane_encoder.py:7-33)."""

    def __init__(self, layers, embedding_size, output_channels, output_size, kernel_size, strides):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv1d(embedding_size, output_channels, kernel_size, strides)])
        width = (10 - kernel_size) // strides + 1          # the reference sizes fc with a (1,2,10) dummy (:20-27)
        self.fc = nn.Linear(output_channels * width, output_size)
        self.stride = strides

    def forward(self, x):
        """x: (R, embedding_size, sample_num) -- the raw ``.view`` of (B,3,10,2) (mgcvae.py:301-302)."""
        stride = getattr(self, "stride", None) or self.convs[0].stride[0]
        n_samples = x.shape[-1]
        packed = self._packed(lambda sd: ops.PackedLane(sd, stride, n_samples))
        return torch_ops.lane_enc_fwd(packed, x.contiguous())


class ViT(_PackedMixin, nn.Module):
    """vit_pytorch==0.33.2 ``ViT`` parameter layout (third-party; parity unpinned)."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool="cls", channels=3,
                 dim_head=64, dropout=0.0, emb_dropout=0.0):
        super().__init__()
        if pool != "cls":
            raise MmtError("only cls pooling is implemented (the reference never passes pool)")
        n_patches = (image_size // patch_size) ** 2
        inner = heads * dim_head
        self.cfg = dict(image_size=image_size, patch_size=patch_size, dim=dim, depth=depth, heads=heads,
                        dim_head=dim_head, mlp_dim=mlp_dim, channels=channels, num_classes=num_classes)
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.Linear(channels * patch_size ** 2, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, n_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)         # (training step only: training.py; the inference kernels are eval-mode)
        blocks = []
        for _ in range(depth):
            attn = nn.Module()
            attn.norm = nn.LayerNorm(dim)
            attn.fn = nn.Module()
            attn.fn.to_qkv = nn.Linear(dim, inner * 3, bias=False)
            attn.fn.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
            ff = nn.Module()
            ff.norm = nn.LayerNorm(dim)
            ff.fn = nn.Module()
            ff.fn.net = nn.Sequential(nn.Linear(dim, mlp_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(mlp_dim, dim),
                                      nn.Dropout(dropout))
            blocks.append(nn.ModuleList([attn, ff]))
        self.transformer = nn.Module()
        self.transformer.layers = nn.ModuleList(blocks)
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))

    def _cfg(self):
        cfg = getattr(self, "cfg", None)
        if cfg is None:       # object unpickled from a vit_pytorch checkpoint: recover the geometry from tensors
            sd = self.state_dict()
            dim = sd["cls_token"].shape[-1]
            inner = sd["transformer.layers.0.0.fn.to_qkv.weight"].shape[0] // 3
            heads = self.transformer.layers[0][0].fn.heads
            depth = len(self.transformer.layers)
            patch_dim = sd["to_patch_embedding.1.weight"].shape[1]
            n_tok = sd["pos_embedding"].shape[1]
            side = int(round((n_tok - 1) ** 0.5))
            patch = int(round((patch_dim / 3) ** 0.5))
            cfg = dict(image_size=side * patch, patch_size=patch, dim=dim, depth=depth, heads=heads,
                       dim_head=inner // heads, mlp_dim=sd["transformer.layers.0.1.fn.net.0.weight"].shape[0],
                       channels=3, num_classes=sd["mlp_head.1.weight"].shape[0])
        return cfg

    def forward(self, img):
        """img is the already rescaled ``map*2-1`` raster, as in the reference's call (mgcvae.py:295-296)."""
        cfg = self._cfg()
        return torch_ops.vit_map_fwd(self._packed(lambda sd: ops.PackedVit(sd, cfg)), img.contiguous(), False)

    def forward_raw(self, maps):
        """maps in [0,1]; ``*2-1`` is applied while the patches are gathered."""
        cfg = self._cfg()
        return torch_ops.vit_map_fwd(self._packed(lambda sd: ops.PackedVit(sd, cfg)), maps.contiguous(), True)
