"""Host-side operator layer: torch tensors in, HIP kernels (through the C ABI) out.

Every function here validates its operands (device, dtype, contiguity, shapes) before the
kernel is launched, launches on torch's current HIP stream and returns torch tensors that were
allocated by the torch caching allocator.  There is deliberately no CPU / ATen implementation of
any op in this file: a CPU tensor or a missing libmmt_amd.so raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import MmtError


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t, name, shape=None, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise MmtError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise MmtError(f"{name}: tensor is on {t.device}; the HIP kernels need device memory (no CPU fallback)")
    if t.dtype != dtype:
        raise MmtError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise MmtError(f"{name}: tensor must be contiguous")
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != ts for s, ts in zip(shape, t.shape)):
            raise MmtError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def _r16(v):
    return (v + 15) & ~15


class _Arena:
    """One flat device buffer holding the packed weights of a module (keeps everything alive)."""

    def __init__(self, device):
        self.device = device
        self.plan = []        # (kind, args) recorded first, so the buffer is allocated once
        self.size = 0
        self.buf = None

    def reserve(self, nfloats):
        off = self.size
        self.size += (nfloats + 63) & ~63          # 256-byte aligned pieces
        return off

    def alloc(self):
        self.buf = torch.zeros(max(self.size, 64), dtype=torch.float32, device=self.device)

    def addr(self, off):
        return self.buf.data_ptr() + 4 * off


class _Packer:
    def __init__(self, device):
        self.arena = _Arena(device)
        self.jobs = []
        self.lib = _lib.load()

    def linear(self, w, b=None, col0=0, k=None, row0=0, n=None):
        """Plan the packing of rows [row0,row0+n) / cols [col0,col0+k) of a (N,K) weight."""
        _chk(w, "weight")
        N, K = w.shape
        n = N - row0 if n is None else n
        k = K - col0 if k is None else k
        woff = self.arena.reserve(self.lib.mmt_packed_linear_floats(n, k))
        boff = self.arena.reserve(_r16(n)) if b is not None else None
        L = _lib.Linear()
        L.n, L.k = n, k
        self.jobs.append(("lin", L, w, b, row0, col0, n, k, woff, boff))
        return L

    def lin4_first(self, w, nparts, part_rows, slice_rows, nslices, tr=False):
        """Plan a copy of a row-split linear in the order of the 4-agent-cluster kernels (mmt_pack_lin4_first); None when
        the geometry is not one they handle.  ``w`` starts at the first row used (a view is fine, row stride = w.stride(0)).
        ``tr``: the matrix to pack is ``w``'s TRANSPOSE (the backward sweep multiplies by W where the forward takes W^T)."""
        k = w.shape[0] if tr else w.shape[1]
        n = self.lib.mmt_lin4_first_floats(nparts, slice_rows, k, nslices)
        if not n or (tr and (nparts != 1 or w.shape[1] != slice_rows * nslices)):
            return None
        holder = [None]
        self.jobs.append(("lin4a", holder, w, nparts, part_rows, slice_rows, nslices, self.arena.reserve(n), bool(tr)))
        return holder

    def lin4_second(self, w, slice_k, nslices, tr=False):
        """Same for a column-split linear (out_proj, linear2): mmt_pack_lin4_second.  ``tr`` as in ``lin4_first``."""
        rows, cols = (w.shape[1], w.shape[0]) if tr else (w.shape[0], w.shape[1])
        n = self.lib.mmt_lin4_second_floats(rows, slice_k, nslices)
        if not n or cols != slice_k * nslices:
            return None
        holder = [None]
        self.jobs.append(("lin4b", holder, w, slice_k, nslices, self.arena.reserve(n), bool(tr)))
        return holder

    def norm(self, g, b):
        goff = self.arena.reserve(_r16(g.numel()))
        boff = self.arena.reserve(_r16(b.numel()))
        Nn = _lib.Norm()
        self.jobs.append(("norm", Nn, g, b, goff, boff))
        return Nn

    def plain(self, t):
        off = self.arena.reserve(t.numel())
        holder = [None]
        self.jobs.append(("plain", holder, t, off))
        return holder

    def finish(self):
        """Allocate the arena, resolve every job to (source pointer, destination pointer, geometry) in a device-resident
        table and fill the arena with one kernel launch."""
        A = self.arena
        A.alloc()
        tab, self._keep = [], []

        def flat(t):          # 1-D contiguous view of the source (parameters are; anything else is copied once and kept)
            v = t.detach().reshape(-1)
            self._keep.append(v)
            return v

        def job(kind, src_ptr, dst_off, total, **kw):
            J = _lib.PackJob()
            J.src, J.dst, J.total, J.kind = src_ptr, A.addr(dst_off), int(total), kind
            for k_, v_ in kw.items():
                setattr(J, k_, int(v_))
            tab.append(J)

        for jb in self.jobs:
            if jb[0] == "lin":
                _, L, w, b, row0, col0, n, k, woff, boff = jb
                job(1, w.data_ptr() + 4 * row0 * w.stride(0), woff, self.lib.mmt_packed_linear_floats(n, k), n=n, k=k,
                    ldw=w.stride(0), col0=col0)
                L.w = A.addr(woff)
                L.b = None
                if b is not None:
                    fb = flat(b)
                    job(0, fb.data_ptr() + 4 * row0, boff, n, n=n)
                    L.b = A.addr(boff)
            elif jb[0] == "lin4a":
                _, holder, w, nparts, part_rows, slice_rows, nslices, off, tr = jb
                k = w.shape[0] if tr else w.shape[1]
                if w.stride(1) != 1:
                    raise MmtError("lin4_first: the source rows must have unit stride")
                job(2, w.data_ptr(), off, self.lib.mmt_lin4_first_floats(nparts, slice_rows, k, nslices),
                    ldw=w.stride(0), nparts=nparts, part_rows=part_rows, slice_rows=slice_rows, k=k, nslices=nslices, tr=int(tr))
                holder[0] = A.addr(off)
            elif jb[0] == "lin4b":
                _, holder, w, slice_k, nslices, off, tr = jb
                rows = w.shape[1] if tr else w.shape[0]
                if w.stride(1) != 1:
                    raise MmtError("lin4_second: the source rows must have unit stride")
                job(3, w.data_ptr(), off, self.lib.mmt_lin4_second_floats(rows, slice_k, nslices), ldw=w.stride(0),
                    n=rows, slice_rows=slice_k, nslices=nslices, tr=int(tr))
                holder[0] = A.addr(off)
            elif jb[0] == "norm":
                _, Nn, g, b, goff, boff = jb
                fg, fb = flat(g), flat(b)
                job(0, fg.data_ptr(), goff, fg.numel(), n=fg.numel())
                job(0, fb.data_ptr(), boff, fb.numel(), n=fb.numel())
                Nn.g, Nn.b = A.addr(goff), A.addr(boff)
            else:
                _, holder, t, off = jb
                ft = flat(t)
                job(0, ft.data_ptr(), off, ft.numel(), n=ft.numel())
                holder[0] = A.addr(off)
        arr = (_lib.PackJob * len(tab))(*tab)
        self._table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(A.device)
        self._njobs = len(tab)
        return self.run()

    def run(self):
        """(Re)fill the arena from the source tensors where they live (same addresses as at finish()): after an in-place
        parameter update this refreshes every packed copy with one launch and no allocation (HIP-graph capturable)."""
        if self._njobs:
            _lib.check(self.lib.mmt_pack_jobs(C.c_void_p(self._table.data_ptr()), self._njobs, _stream()), "mmt_pack_jobs")
        return self.arena


def _copy_struct(dst, src):
    C.memmove(C.byref(dst), C.byref(src), C.sizeof(src))


def _mlp_desc(pk, sd, prefix):
    """Mlp (MLP.py) regression form from state_dict names ``layer_seq.lmlp_i.{0,1}.*`` / ``layer_seq.out_mlp.*``."""
    M = _lib.Mlp()
    i = 0
    pend = []
    while f"{prefix}layer_seq.lmlp_{i}.0.weight" in sd:
        if i >= _lib.MMT_MAX_MLP_HIDDEN:
            raise MmtError(f"Mlp with more than {_lib.MMT_MAX_MLP_HIDDEN} hidden layers is not supported")
        p = f"{prefix}layer_seq.lmlp_{i}."
        pend.append((pk.linear(sd[p + "0.weight"], sd[p + "0.bias"]), pk.norm(sd[p + "1.weight"], sd[p + "1.bias"])))
        i += 1
    if f"{prefix}layer_seq.out_mlp.weight" not in sd:
        raise MmtError("only regression-mode Mlp can run inside the decoder kernel")
    out = pk.linear(sd[f"{prefix}layer_seq.out_mlp.weight"], sd[f"{prefix}layer_seq.out_mlp.bias"])
    return M, pend, out


def _fill_mlp(M, pend, out):
    M.n_hidden = len(pend)
    for i, (lin, nrm) in enumerate(pend):
        _copy_struct(M.hidden[i], lin)
        _copy_struct(M.hidden_ln[i], nrm)
    _copy_struct(M.out, out)


def _version_key(tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors)


class PackedEncoder:
    """Device-side descriptor of an Encoder (Transformer_model.py:11-59) built from its state_dict."""

    def __init__(self, sd: dict, nhead: int):
        w_in = sd["input_fc.weight"]
        dev = w_in.device
        d, n_in = w_in.shape
        pk = _Packer(dev)
        E = _lib.Encoder()
        E.d_model, E.nhead, E.n_in = d, nhead, n_in
        lin_in = pk.linear(w_in, sd["input_fc.bias"])
        pe = pk.plain(sd["pos_encoder.pe"][:64])     # T <= 32 rows are ever read
        layers = []
        L = 0
        while f"transformer_encoder.layers.{L}.linear1.weight" in sd:
            p = f"transformer_encoder.layers.{L}."
            layers.append(dict(
                qkv=pk.linear(sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]),
                out=pk.linear(sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"]),
                ff1=pk.linear(sd[p + "linear1.weight"], sd[p + "linear1.bias"]),
                ff2=pk.linear(sd[p + "linear2.weight"], sd[p + "linear2.bias"]),
                ln1=pk.norm(sd[p + "norm1.weight"], sd[p + "norm1.bias"]),
                ln2=pk.norm(sd[p + "norm2.weight"], sd[p + "norm2.bias"])))
            L += 1
        if L == 0 or L > _lib.MMT_MAX_LAYERS:
            raise MmtError(f"encoder with {L} layers is not supported (1..{_lib.MMT_MAX_LAYERS})")
        E.nlayers = L
        E.d_ff = sd["transformer_encoder.layers.0.linear1.weight"].shape[0]
        self.arena = pk.finish()
        _copy_struct(E.input_fc, lin_in)
        E.pe = pe[0]
        for i, lay in enumerate(layers):
            for k, v in lay.items():
                _copy_struct(getattr(E.layer[i], k), v)
        self.desc = E
        self.d_model = d
        self.n_in = n_in


class PackedDecoder:
    """Trajectory_Decoder + head Mlp (+ optional map / memory fusion Mlps) descriptor."""

    def __init__(self, sd: dict, nhead: int, head_sd: dict, map_fusion_sd: dict | None = None,
                 mem_fusion_sd: dict | None = None, mem_fusion_is_mix_fc: bool = False):
        w_in = sd["input_fc.weight"]
        dev = w_in.device
        d, n_in = w_in.shape
        pk = _Packer(dev)
        D = _lib.Decoder()
        D.d_model, D.nhead, D.n_in = d, nhead, n_in
        D.n_out = sd["output_fc.weight"].shape[0]
        lin_in = pk.linear(w_in, sd["input_fc.bias"])
        lin_out = pk.linear(sd["output_fc.weight"], sd["output_fc.bias"])
        pe = pk.plain(sd["pos_encoder.pe"][:64])
        layers, blocks = [], []
        L = 0
        while f"transformer_decoder.layers.{L}.linear1.weight" in sd:
            p = f"transformer_decoder.layers.{L}."
            cw, cb = sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"]
            layers.append(dict(
                qkv=pk.linear(sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]),
                out=pk.linear(sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"]),
                cq=pk.linear(cw, cb, row0=0, n=d),
                ckv=pk.linear(cw, cb, row0=d, n=2 * d),
                cout=pk.linear(sd[p + "multihead_attn.out_proj.weight"], sd[p + "multihead_attn.out_proj.bias"]),
                ff1=pk.linear(sd[p + "linear1.weight"], sd[p + "linear1.bias"]),
                ff2=pk.linear(sd[p + "linear2.weight"], sd[p + "linear2.bias"]),
                ln1=pk.norm(sd[p + "norm1.weight"], sd[p + "norm1.bias"]),
                ln2=pk.norm(sd[p + "norm2.weight"], sd[p + "norm2.bias"]),
                ln3=pk.norm(sd[p + "norm3.weight"], sd[p + "norm3.bias"])))
            f = sd[p + "linear1.weight"].shape[0]
            if d % nhead == 0 and f % nhead == 0:
                dh, fs = d // nhead, f // nhead
                blocks.append(dict(
                    qkv4=pk.lin4_first(sd[p + "self_attn.in_proj_weight"], 3, d, dh, nhead),
                    out4=pk.lin4_second(sd[p + "self_attn.out_proj.weight"], dh, nhead),
                    cq4=pk.lin4_first(cw[:d], 1, 0, dh, nhead),
                    cout4=pk.lin4_second(sd[p + "multihead_attn.out_proj.weight"], dh, nhead),
                    ff14=pk.lin4_first(sd[p + "linear1.weight"], 1, 0, fs, nhead),
                    ff24=pk.lin4_second(sd[p + "linear2.weight"], fs, nhead)))
            else:
                blocks.append({})
            L += 1
        if L == 0 or L > _lib.MMT_MAX_LAYERS:
            raise MmtError(f"decoder with {L} layers is not supported (1..{_lib.MMT_MAX_LAYERS})")
        D.nlayers = L
        D.d_ff = sd["transformer_decoder.layers.0.linear1.weight"].shape[0]
        head = _mlp_desc(pk, head_sd, "") if head_sd is not None else None
        mapf = _mlp_desc(pk, map_fusion_sd, "") if map_fusion_sd is not None else None
        memf = None
        if mem_fusion_sd is not None:
            if mem_fusion_is_mix_fc:   # Trajectory_Decoder.mix_fc is a bare Linear (Transformer_model.py:166,189-191)
                memf = (_lib.Mlp(), [], pk.linear(mem_fusion_sd["weight"], mem_fusion_sd["bias"]))
            else:
                memf = _mlp_desc(pk, mem_fusion_sd, "")
        self.arena = pk.finish()
        self._packer = pk
        _copy_struct(D.input_fc, lin_in)
        _copy_struct(D.output_fc, lin_out)
        D.pe = pe[0]
        for i, lay in enumerate(layers):
            for k, v in lay.items():
                _copy_struct(getattr(D.layer[i], k), v)
            for k, h in blocks[i].items():
                setattr(D.layer[i], k, h[0] if h is not None else None)
        if head is not None:
            _fill_mlp(D.head, head[1], head[2])
        D.has_map_fusion = int(mapf is not None)
        if mapf is not None:
            _fill_mlp(D.map_fusion, mapf[1], mapf[2])
        D.has_mem_fusion = int(memf is not None)
        if memf is not None:
            _fill_mlp(D.mem_fusion, memf[1], memf[2])
        self.has_head = head is not None
        self.desc = D
        self.d_model, self.n_in, self.n_out = d, n_in, D.n_out


    def repack(self):
        """Refresh the packed copies from the parameter tensors this descriptor was built from (same addresses)."""
        self._packer.run()


class PackedCnn:
    """CNNMapEncoder (map_encoder.py:6-36) descriptor: plain conv / fc weights copied into one arena."""

    def __init__(self, sd: dict, strides, in_hw):
        pk = _Packer(sd["fc.weight"].device)
        N = _lib.Cnn()
        holders = []
        i = 0
        while f"convs.{i}.weight" in sd:
            if i >= _lib.MMT_MAX_CONV:
                raise MmtError(f"more than {_lib.MMT_MAX_CONV} conv layers are not supported")
            w = sd[f"convs.{i}.weight"]
            if w.shape[2] != w.shape[3]:
                raise MmtError("only square conv kernels are supported")
            N.out_ch[i], N.ksize[i], N.stride[i] = w.shape[0], w.shape[2], int(strides[i])
            if i == 0:
                N.in_ch = w.shape[1]
            holders.append((pk.plain(w), pk.plain(sd[f"convs.{i}.bias"])))
            i += 1
        N.nconv = i
        N.in_h, N.in_w = int(in_hw[0]), int(in_hw[1])
        fw, fb = pk.plain(sd["fc.weight"]), pk.plain(sd["fc.bias"])
        N.n_out = sd["fc.weight"].shape[0]
        self.flat = sd["fc.weight"].shape[1]
        self.arena = pk.finish()
        for j, (hw, hb) in enumerate(holders):
            N.conv_w[j], N.conv_b[j] = hw[0], hb[0]
        N.fc_w, N.fc_b = fw[0], fb[0]
        # MFMA lane-ordered copies of the conv weights (used when the architecture is the reference's)
        lib = _lib.load()
        self.packed_convs = []
        for j in range(N.nconv):
            w = sd[f"convs.{j}.weight"].detach().contiguous()
            dst = torch.empty(lib.mmt_packed_conv_floats(w.shape[0], w.shape[1], w.shape[2]), dtype=torch.float32,
                              device=w.device)
            _lib.check(lib.mmt_pack_conv(_ptr(w), w.shape[0], w.shape[1], w.shape[2], _ptr(dst), _stream()), "mmt_pack_conv")
            self.packed_convs.append(dst)
            N.conv_wp[j] = dst.data_ptr()
        self.desc = N


class PackedVit:
    """vit_pytorch 0.33.2 ``ViT`` descriptor (state_dict names of that package; parity unpinned)."""

    HEADS_PER_GROUP = 2
    PANEL_K = 1024

    def __init__(self, sd: dict, cfg: dict):
        dev = sd["cls_token"].device
        pk = _Packer(dev)
        V = _lib.Vit()
        dim, heads, dh = cfg["dim"], cfg["heads"], cfg["dim_head"]
        inner = heads * dh
        # as many heads per q|k|v group as the LDS tile holds (fewer, wider linears and barriers per block): rows x (residual + LN
        # + scratch of 3*gw columns) within 160 KiB; 17 rows when the kernel's 16-patch form applies, else 32
        ntok = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
        rows, pad = (17 if ntok == 17 else 32), 8
        hg = 1
        for cand in range(heads, 0, -1):
            scr = max(3 * cand * dh, cfg["mlp_dim"], self.PANEL_K, (cfg["num_classes"] + 15) & ~15) + pad
            need = max(rows * (2 * (dim + pad) + scr), ((17 + 32) * (dim + pad) + 16 * scr) if rows == 17 else 0) * 4
            if heads % cand == 0 and need <= 160 * 1024:
                hg = cand
                break
        gw = hg * dh
        V.image, V.patch, V.channels = cfg["image_size"], cfg["patch_size"], cfg["channels"]
        V.dim, V.depth, V.heads, V.dim_head, V.mlp_dim = dim, cfg["depth"], heads, dh, cfg["mlp_dim"]
        V.n_out, V.heads_per_group, V.panel_k = cfg["num_classes"], hg, self.PANEL_K
        if V.depth > _lib.MMT_MAX_LAYERS:
            raise MmtError(f"ViT depth {V.depth} > {_lib.MMT_MAX_LAYERS}")
        wp, bp = sd["to_patch_embedding.1.weight"], sd["to_patch_embedding.1.bias"]
        patch_dim = wp.shape[1]
        npanel = (patch_dim + self.PANEL_K - 1) // self.PANEL_K
        # grouped sub-matrices must be back to back in the arena: their biases are packed separately
        panels = [pk.linear(wp, None, col0=i * self.PANEL_K, k=min(self.PANEL_K, patch_dim - i * self.PANEL_K))
                  for i in range(npanel)]
        patch_bias = pk.plain(bp)
        cls, pos = pk.plain(sd["cls_token"]), pk.plain(sd["pos_embedding"])
        blocks = []
        for i in range(V.depth):
            a, f = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1."
            wq = sd[a + "fn.to_qkv.weight"]
            qkv = [pk.linear(wq, None, row0=t * inner + g * gw, n=gw) for g in range(heads // hg) for t in range(3)]
            wo, bo = sd[a + "fn.to_out.0.weight"], sd[a + "fn.to_out.0.bias"]
            outs = [pk.linear(wo, None, col0=g * gw, k=gw) for g in range(heads // hg)]
            blocks.append(dict(
                out_bias=pk.plain(bo),
                ln1=pk.norm(sd[a + "norm.weight"], sd[a + "norm.bias"]), qkv=qkv, out=outs,
                ln2=pk.norm(sd[f + "norm.weight"], sd[f + "norm.bias"]),
                ff1=pk.linear(sd[f + "fn.net.0.weight"], sd[f + "fn.net.0.bias"]),
                ff2=pk.linear(sd[f + "fn.net.3.weight"], sd[f + "fn.net.3.bias"])))
        hln = pk.norm(sd["mlp_head.0.weight"], sd["mlp_head.0.bias"])
        hfc = pk.linear(sd["mlp_head.1.weight"], sd["mlp_head.1.bias"])
        self.arena = pk.finish()

        def contiguous(lins, what):
            step = 4 * (((_r16(lins[0].n) * _r16(lins[0].k)) + 63) & ~63)
            for a_, b_ in zip(lins, lins[1:]):
                if b_.w - a_.w != step:
                    raise MmtError(f"internal: {what} sub-matrices are not contiguous in the arena")
            if step != 4 * _r16(lins[0].n) * _r16(lins[0].k):
                raise MmtError(f"internal: {what} packed size is not 64-float aligned")

        contiguous(panels, "panel")
        _copy_struct(V.patch_fc, panels[0])
        V.patch_fc.k = patch_dim
        V.patch_fc.b = patch_bias[0]
        V.cls_token, V.pos_embedding = cls[0], pos[0]
        for i, blk in enumerate(blocks):
            contiguous(blk["qkv"], "qkv")
            contiguous(blk["out"], "out")
            B_ = V.block[i]
            _copy_struct(B_.ln1, blk["ln1"])
            _copy_struct(B_.ln2, blk["ln2"])
            _copy_struct(B_.qkv, blk["qkv"][0])
            B_.qkv.k = dim
            _copy_struct(B_.out, blk["out"][0])
            B_.out.n = dim
            B_.out.b = blk["out_bias"][0]
            _copy_struct(B_.ff1, blk["ff1"])
            _copy_struct(B_.ff2, blk["ff2"])
        _copy_struct(V.head_ln, hln)
        _copy_struct(V.head_fc, hfc)
        self.desc = V


class PackedLane:
    """Lane_Encoder (Lane_encoder.py:7-33) descriptor."""

    def __init__(self, sd: dict, stride: int, n_samples: int):
        pk = _Packer(sd["fc.weight"].device)
        N = _lib.Lane()
        w = sd["convs.0.weight"]
        N.out_ch, N.in_ch, N.ksize = w.shape
        N.stride, N.n_samples = int(stride), int(n_samples)
        N.n_out = sd["fc.weight"].shape[0]
        hs = [pk.plain(w), pk.plain(sd["convs.0.bias"]), pk.plain(sd["fc.weight"]), pk.plain(sd["fc.bias"])]
        self.arena = pk.finish()
        N.conv_w, N.conv_b, N.fc_w, N.fc_b = (h[0] for h in hs)
        self.desc = N
        width = (n_samples - N.ksize) // N.stride + 1
        if N.out_ch * width != sd["fc.weight"].shape[1]:
            raise MmtError("lane encoder: fc input size does not match conv output")


# ------------------------------------------------------------------------------------------ ops
def encoder_forward(enc: PackedEncoder, x_st: torch.Tensor, want_mask: bool = True):
    """Encoder.forward incl. the causal / key-padding masks -> (memory (B,T,d), pad_mask (B,T) uint8)."""
    _chk(x_st, "x_st", (None, None, enc.n_in))
    B, T, _ = x_st.shape
    if B == 0:
        raise MmtError("x_st: empty batch")
    memory = torch.empty(B, T, enc.d_model, dtype=torch.float32, device=x_st.device)
    pad = torch.empty(B, T, dtype=torch.uint8, device=x_st.device) if want_mask else None
    lib = _lib.load()
    ws = _encoder_workspace(lib.mmt_encoder_workspace_floats(C.byref(enc.desc), B, T), x_st.device) if B <= 256 else None
    _lib.check(lib.mmt_encoder_forward_ws(C.byref(enc.desc), _ptr(x_st), B, T, _ptr(memory), _ptr(pad), _ptr(ws), _stream()),
               "mmt_encoder_forward_ws")
    return memory, pad


def decoder_select(impl: str = "auto"):
    """Choose the kernel family behind ``decoder_forward``: "auto" | "persistent" | "cluster" | "cluster_launches"
    (include/mmt_amd.h)."""
    table = {"auto": _lib.MMT_DECODER_AUTO, "persistent": _lib.MMT_DECODER_PERSISTENT, "cluster": _lib.MMT_DECODER_CLUSTER,
             "cluster_launches": _lib.MMT_DECODER_CLUSTER_LAUNCHES, "cluster_fused": _lib.MMT_DECODER_CLUSTER_FUSED,
             "cluster_pipeline": _lib.MMT_DECODER_CLUSTER_PIPELINE}
    if impl not in table:
        raise MmtError(f"decoder_select: unknown implementation {impl!r}")
    _lib.check(_lib.load().mmt_decoder_select(table[impl]), "mmt_decoder_select")
    _bump_form_generation()


DECODER_PATHS = {0: "none", 1: "mmt_decoder_kernel (persistent decoder, one launch)",
                 2: "mmt_dec_phase_kernel per phase (16-row clusters) + mmt_dec_ckv_kernel",
                 3: "mmt_dec_phase4_kernel per phase (4-agent clusters) + mmt_dec_ckv_kernel + mmt_dec_phase_kernel (final head)",
                 4: "mmt_dec_fused4_kernel (every phase in one launch) + mmt_dec_ckv_kernel + mmt_dec_phase_kernel (final head)",
                 5: "mmt_dec_pipe4_kernel (weight-stationary pipeline, one launch) + mmt_dec_ckv_kernel + mmt_dec_phase_kernel (final head)"}


def decoder_last_path():
    """Kernel family the last decoder_forward of this thread actually ran (code, description)."""
    code = int(_lib.load().mmt_decoder_last_path())
    return code, DECODER_PATHS.get(code, "?")


_form_generation = [0]


def _bump_form_generation():
    _form_generation[0] += 1


def decoder_form_generation():
    """Moves whenever the kernel form behind decoder_forward / encoder_forward may have changed (decoder_select, a one-launch form
    disabled after a time-out): a HIP graph captured under an older value bakes in a form that must not be replayed."""
    return _form_generation[0]


def decoder_status(stream=None):
    """Synchronise and raise MmtError if a one-launch decode on the current device timed out since the last check (its outputs
    are NaN; the library then keeps that form disabled on the device, so re-issuing the call succeeds)."""
    try:
        _lib.check(_lib.load().mmt_decoder_status(_stream() if stream is None else stream), "mmt_decoder_status")
    except MmtError:
        _enc_ws_cache.clear()      # (a timed-out split encoder leaves its progress words inconsistent: fresh, zeroed buffers next time)
        _bump_form_generation()
        raise


def checked(fn):
    """Run ``fn`` (a callable that launches decodes and returns device tensors), then check the decoder status at the caller's
    synchronisation point; on a timed-out one-launch decode re-issue ``fn`` once (it then runs launch-per-phase)."""
    out = fn()
    try:
        decoder_status()
    except MmtError as e:
        import warnings
        warnings.warn(f"{e}; re-issuing", RuntimeWarning)
        out = fn()
        decoder_status()
    return out


_ws_cache = {}


def _workspace(nfloats, device):
    # one buffer per (device, stream): kernels of two streams (the map encoder runs beside the history encoder) never share scratch
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(int(nfloats * 1.25) + 1024, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


_enc_ws_cache = {}


def _encoder_workspace(nfloats, device):
    """Exchange buffers + progress words of the two-workgroups-per-agent encoder: ZEROED when allocated (the kernel keeps the words
    consistent from call to call), one per (device, stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _enc_ws_cache.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.zeros(int(nfloats) + 1024, dtype=torch.float32, device=device)
        _enc_ws_cache[key] = buf
    return buf


def decoder_dropout_supported(dec: PackedDecoder, B, T, ph):
    """Whether ``decoder_forward(..., dropout=...)`` can run this shape (the 4-agent-cluster kernels are the ones that read masks)."""
    return bool(_lib.load().mmt_decoder_dropout_supported(C.byref(dec.desc), int(B), int(T), int(ph)))


def _dropout_desc(dec, sites, B, T, ph):
    """mmt_decoder_dropout from {site name: (uint8 keep tensor, scale) | None} with the training step's site names and shapes
    ('dec.pe', 'dec{l}.attn1|drop1|attn2|drop2|ff|drop3')."""
    D = _lib.DecoderDropout()
    H, d, f = dec.desc.nhead, dec.desc.d_model, dec.desc.d_ff

    def put(name, shape):
        v = sites.get(name)
        if v is None:
            return 0, 0.0
        keep, scale = v
        if keep.dtype != torch.uint8 or not keep.is_cuda or not keep.is_contiguous() or keep.numel() != shape:
            raise MmtError(f"dropout site {name}: expected a contiguous device uint8 tensor of {shape} elements")
        return keep.data_ptr(), float(scale)

    D.pe, D.pe_scale = put("dec.pe", ph * B * d)
    for l in range(dec.desc.nlayers):
        for kind, n in (("attn1", B * H * ph * ph), ("drop1", ph * B * d), ("attn2", B * H * ph * T), ("drop2", ph * B * d),
                        ("ff", ph * B * f), ("drop3", ph * B * d)):
            ptr, sc = put(f"dec{l}.{kind}", n)
            getattr(D, kind)[l] = ptr
            getattr(D, kind + "_scale")[l] = sc
    return D


def decoder_workspace_floats(dec: PackedDecoder, B, T, ph):
    return int(_lib.load().mmt_decoder_workspace_floats(C.byref(dec.desc), B, T, ph))


def decoder_stash_layout(dec: PackedDecoder, B, T, ph, map_dim=0):
    """Offsets (floats) of the training stash that ``decoder_forward(..., stash=)`` fills (mmt_decoder_stash_layout), or None where
    the geometry has no stash-keeping decode."""
    lay = _lib.DecoderStashLayout()
    if _lib.load().mmt_decoder_stash_layout_of(C.byref(dec.desc), map_dim, B, T, ph, C.byref(lay)) != 0:
        return None
    return {n: int(getattr(lay, n)) for n in _lib.DecoderStashLayout.FIELDS}


def decoder_stash_supported(dec: PackedDecoder, B, T, ph, with_dropout=False, map_dim=0):
    """Whether the next decode can keep the stash (static geometry AND the device's present state: mmt_decoder_stash_supported)."""
    return bool(_lib.load().mmt_decoder_stash_supported(C.byref(dec.desc), map_dim, B, T, ph, int(bool(with_dropout))))


def decoder_forward(dec: PackedDecoder, memory, pad_mask, start_pos=None, ph=12, map_feat=None, mem_feat=None,
                    tgt=None, want_h=False, pred_out=None, dropout=None, stash=None, workspace=None):
    """Autoregressive (tgt=None) or teacher-forced decode.  Returns (pred_st (B,ph,n_in) | None, h (B,ph,n_out) | None).
    ``pred_out``: caller-owned (B,ph,n_in) buffer for the autoregressive result (no allocation in this call).
    ``dropout``: {site: (keep, scale) | None} -- the training step's decode with its keep masks (autoregressive only).
    ``stash``: flat float32 buffer laid out as ``decoder_stash_layout`` says -- the decode also keeps what the backward reads
    (mmt_decoder_forward_stash; autoregressive only, raises where ``decoder_stash_supported`` is False).
    ``workspace``: caller-owned buffer of >= ``decoder_workspace_floats`` floats instead of the per-stream cached one (its first
    nlayers * B * T * 2 * d_model floats receive the cross-attention K/V of every layer, [layer][b * T + t][k | v])."""
    _chk(memory, "memory", (None, None, dec.d_model))
    B, T, _ = memory.shape
    if pad_mask is not None:
        _chk(pad_mask, "pad_mask", (B, T), torch.uint8)
    if tgt is None:
        _chk(start_pos, "start_pos", (B, dec.n_in))
        if not dec.has_head:
            raise MmtError("autoregressive decoding needs the decoder/MLP head")
    else:
        _chk(tgt, "tgt", (B, None, dec.n_in))
        ph = tgt.shape[1]
    md = mf = 0
    if map_feat is not None:
        _chk(map_feat, "map_feat", (B, None))
        md = map_feat.shape[1]
    if mem_feat is not None:
        _chk(mem_feat, "mem_feat", (B, None))
        mf = mem_feat.shape[1]
    dev = memory.device
    pred = None
    if tgt is None:
        if pred_out is not None:
            _chk(pred_out, "pred_out", (B, ph, dec.n_in))
            pred = pred_out
        else:
            pred = torch.empty(B, ph, dec.n_in, dtype=torch.float32, device=dev)
    h = torch.empty(B, ph, dec.n_out, dtype=torch.float32, device=dev) if (want_h or tgt is not None) else None
    lib = _lib.load()
    need = lib.mmt_decoder_workspace_floats(C.byref(dec.desc), B, T, ph)
    if workspace is not None:
        _chk(workspace, "workspace", (None,))
        if workspace.numel() < need:
            raise MmtError(f"decoder_forward: workspace of {workspace.numel()} floats, need {need}")
        ws = workspace
    else:
        ws = _workspace(need, dev)
    if stash is not None:
        if tgt is not None or want_h:
            raise MmtError("decoder_forward: the stash-keeping decode is autoregressive")
        _chk(stash, "stash", (None,))
        D = _dropout_desc(dec, dropout, B, T, ph) if dropout is not None else None
        _lib.check(lib.mmt_decoder_forward_stash(C.byref(dec.desc), _ptr(memory), _ptr(pad_mask), _ptr(start_pos), _ptr(map_feat), md,
                                                 _ptr(mem_feat), mf, B, T, ph, _ptr(pred), _ptr(ws),
                                                 C.byref(D) if D is not None else None, _ptr(stash), stash.numel(), _stream()),
                   "mmt_decoder_forward_stash")
        return pred, None
    if dropout is not None:
        if tgt is not None or want_h:
            raise MmtError("decoder_forward: keep masks go with the autoregressive decode only")
        D = _dropout_desc(dec, dropout, B, T, ph)
        _lib.check(lib.mmt_decoder_forward_dropout(C.byref(dec.desc), _ptr(memory), _ptr(pad_mask), _ptr(start_pos), _ptr(map_feat),
                                                   md, _ptr(mem_feat), mf, B, T, ph, _ptr(pred), _ptr(ws), C.byref(D), _stream()),
                   "mmt_decoder_forward_dropout")
        return pred, None
    _lib.check(lib.mmt_decoder_forward(C.byref(dec.desc), _ptr(memory), _ptr(pad_mask), _ptr(start_pos), _ptr(map_feat),
                                       md, _ptr(mem_feat), mf, _ptr(tgt), B, T, ph, _ptr(pred), _ptr(h), _ptr(ws),
                                       _stream()), "mmt_decoder_forward")
    return pred, h


def cnn_forward(cnn: PackedCnn, maps: torch.Tensor, input_affine: bool = True):
    """CNNMapEncoder.forward: maps (B,C,H,W) -> (B,n_out); input_affine applies map*2-1 inside the kernel."""
    N = cnn.desc
    N.input_affine = int(bool(input_affine))
    _chk(maps, "map", (None, N.in_ch, N.in_h, N.in_w))
    B = maps.shape[0]
    out = torch.empty(B, N.n_out, dtype=torch.float32, device=maps.device)
    lib = _lib.load()
    _lib.check(lib.mmt_cnn_forward(C.byref(N), _ptr(maps), B, _ptr(out), None, _stream()), "mmt_cnn_forward")
    return out


def vit_forward(vit: PackedVit, maps: torch.Tensor, input_affine: bool = True):
    """ViT.forward: maps (B,C,S,S) -> (B,num_classes); input_affine applies map*2-1 inside the kernel."""
    V = vit.desc
    V.input_affine = int(bool(input_affine))
    _chk(maps, "map", (None, V.channels, V.image, V.image))
    B = maps.shape[0]
    out = torch.empty(B, V.n_out, dtype=torch.float32, device=maps.device)
    lib = _lib.load()
    _lib.check(lib.mmt_vit_forward(C.byref(V), _ptr(maps), B, _ptr(out), _stream()), "mmt_vit_forward")
    return out


def lane_forward(lane: PackedLane, lanes: torch.Tensor):
    """Lane_Encoder.forward on rows of (in_ch*n_samples) floats: (R, in_ch, n_samples) -> (R, n_out)."""
    N = lane.desc
    _chk(lanes, "lanes", (None, N.in_ch, N.n_samples))
    R = lanes.shape[0]
    out = torch.empty(R, N.n_out, dtype=torch.float32, device=lanes.device)
    lib = _lib.load()
    _lib.check(lib.mmt_lane_forward(C.byref(N), _ptr(lanes), R, _ptr(out), _stream()), "mmt_lane_forward")
    return out


def finalize(pred_st, node_pos=None, gt=None, metres=True, want_pred=True, want_metrics=False, want_loss=False):
    """pred_st*80+node_pos, ADE/FDE and the summed per-agent mean L2 in one pass over (B,ph,2)."""
    _chk(pred_st, "pred_st", (None, None, 2))
    B, ph, _ = pred_st.shape
    dev = pred_st.device
    if node_pos is not None:
        _chk(node_pos, "node_pos", (B, 2))
    if gt is not None:
        _chk(gt, "gt", (B, ph, 2))
    pred_m = torch.empty_like(pred_st) if want_pred else None
    ade = torch.empty(B, dtype=torch.float32, device=dev) if want_metrics else None
    fde = torch.empty(B, dtype=torch.float32, device=dev) if want_metrics else None
    loss = torch.zeros(1, dtype=torch.float32, device=dev) if want_loss else None
    lib = _lib.load()
    _lib.check(lib.mmt_finalize(_ptr(pred_st), _ptr(node_pos), _ptr(gt), int(bool(metres)), B, ph, _ptr(pred_m), _ptr(ade),
                                _ptr(fde), _ptr(loss), _stream()), "mmt_finalize")
    return pred_m, ade, fde, loss
