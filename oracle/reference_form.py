"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of the reference hot path *in reference form*: stock
``torch.nn.Transformer*`` containers and the step-by-step autoregressive loop
that recomputes the whole decoder on a growing target every step.  It exists
for two reasons only:

  * tests compare the HIP path with it (and it is itself pinned against golden
    vectors generated from the real reference modules by
    ``tests/golden/make_golden.py``);
  * ``bench.py``'s ``cpu_baseline`` leg times it on the GPU box's host cores
    (the reference's own files cannot travel there).

Nothing under ``multimodal_transformer_amd/`` may import this module.

What each piece follows in the reference (paths relative to /root/reference):
  HistoryEncoder      Transformer/nn_component/Transformer_model.py:11-59
  SinusoidTable       Transformer/nn_component/Transformer_model.py:112-129
  TrajectoryDecoder   Transformer/nn_component/Transformer_model.py:62-109,155-199
  MlpStack            Transformer/nn_component/MLP.py:6-30
  MapCNN              Transformer/nn_component/map_encoder.py:6-36
  LaneConv            Transformer/nn_component/Lane_encoder.py:7-33
  MapViT              third-party vit_pytorch==0.33.2 (requirements-ci.txt:15), NOT in the
                      reference tree: restated from its published architecture -> parity UNPINNED
  build_modules       Transformer/model/mgcvae.py:77-212 (registry keys and ctor args)
  encode / ar_decode / predict / train_loss / eval_loss
                      Transformer/model/mgcvae.py:231-308, 375-449, 559-773
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

STD = 80.0  # environment/environment.py:61 divides every state dim by the attention radius


def causal_mask(n: int, device=None) -> torch.Tensor:
    """0 on/below the diagonal, -inf above (model_utils.py:141-145)."""
    m = torch.full((n, n), float("-inf"), device=device)
    return torch.triu(m, diagonal=1)


def padding_mask(x_st: torch.Tensor) -> torch.Tensor:
    """True where the history step is padding: NaN in feature column 1 (model_utils.py:148-151)."""
    return torch.isnan(x_st)[:, :, 1]


class SinusoidTable(nn.Module):
    def __init__(self, d_model: int, p_drop: float, max_len: int = 5000):
        super().__init__()
        self.dropout = nn.Dropout(p_drop)
        pos = torch.arange(0, max_len).unsqueeze(1).float()
        freq = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        table = torch.zeros(max_len, d_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table)

    def forward(self, x):
        return self.dropout(x + self.pe[: x.shape[-2]])


class HistoryEncoder(nn.Module):
    def __init__(self, ninp=8, ntimestep=9, in_dim=128, nhead=2, fdim=256, nlayers=6, noutput=32,
                 dropout=0.2):
        super().__init__()
        self.input_fc = nn.Linear(ninp, in_dim)
        self.pos_encoder = SinusoidTable(in_dim, dropout)
        layer = nn.TransformerEncoderLayer(in_dim, nhead, fdim, dropout, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(layer, nlayers)
        self.output_fc = nn.Linear(in_dim, noutput)  # present in checkpoints, unused by forward

    def forward(self, src, src_mask, src_key_padding_mask):
        h = self.pos_encoder(self.input_fc(torch.nan_to_num(src, nan=0)))
        return self.transformer_encoder(h, mask=src_mask, src_key_padding_mask=src_key_padding_mask)


class TrajectoryDecoder(nn.Module):
    def __init__(self, tgt_inp, lane_inp, in_dim, nhead, fdim, nlayers, noutput, dropout=0.2):
        super().__init__()
        self.input_fc = nn.Linear(tgt_inp, in_dim)
        self.pos_encoder = SinusoidTable(in_dim, dropout)
        layer = nn.TransformerDecoderLayer(in_dim, nhead, fdim, dropout, batch_first=True)
        self.transformer_decoder = nn.TransformerDecoder(layer, nlayers)
        self.output_fc = nn.Linear(in_dim, noutput)
        self.mix_fc = nn.Linear(lane_inp + in_dim, in_dim)

    def forward(self, tgt, memory, tgt_mask, memory_mask, memory_key_padding_mask, lane_feature=None):
        if lane_feature is not None:
            memory = self.mix_fc(torch.cat([memory, lane_feature], dim=-1))
        # memory_mask is accepted and dropped, exactly like the reference (:192-197)
        h = self.transformer_decoder(self.pos_encoder(self.input_fc(tgt)), memory, tgt_mask=tgt_mask,
                                     memory_key_padding_mask=memory_key_padding_mask)
        return self.output_fc(h)


class MlpStack(nn.Module):
    def __init__(self, in_channels, output_size, layer_num, mode):
        super().__init__()
        self.layer_seq = nn.Sequential()
        for i in range(layer_num - 1):
            self.layer_seq.add_module(
                f"lmlp_{i}", nn.Sequential(nn.Linear(in_channels, in_channels), nn.LayerNorm(in_channels), nn.ReLU()))
        out = nn.Linear(in_channels, output_size)
        if mode == "classification":
            # nn.Softmax() with dim=None: on a 3-D input torch picks dim 0 (the batch!) -- MLP.py:21
            self.layer_seq.add_module("out_mlp", nn.Sequential(out, nn.Softmax()))
        else:
            self.layer_seq.add_module("out_mlp", out)

    def forward(self, x):
        return self.layer_seq(x)


class MapCNN(nn.Module):
    def __init__(self, map_channels, hidden_channels, output_size, masks, strides, patch_size):
        super().__init__()
        side_x = patch_size[0] + patch_size[2]
        side_y = patch_size[1] + patch_size[3]
        chans = [map_channels] + list(hidden_channels)
        self.convs = nn.ModuleList(
            nn.Conv2d(chans[i], chans[i + 1], masks[i], stride=strides[i]) for i in range(len(hidden_channels)))
        with torch.no_grad():
            probe = torch.zeros(1, map_channels, side_x, side_y)
            for c in self.convs:
                probe = c(probe)
        self.fc = nn.Linear(probe.numel(), output_size)

    def forward(self, x, training=False):
        for c in self.convs:
            x = F.leaky_relu(c(x), 0.2)
        return self.fc(torch.flatten(x, start_dim=1))


class LaneConv(nn.Module):
    def __init__(self, layers, embedding_size, output_channels, output_size, kernel_size, strides, sample_num=10):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv1d(embedding_size, output_channels, kernel_size, strides)])
        width = (sample_num - kernel_size) // strides + 1
        self.fc = nn.Linear(output_channels * width, output_size)

    def forward(self, x):
        for c in self.convs:
            x = F.leaky_relu(c(x), 0.2)
        return self.fc(torch.flatten(x, start_dim=1))


# --------------------------------------------------------------------------- ViT (unpinned)
class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x):
        return self.fn(self.norm(x))


class _VitFF(nn.Module):
    def __init__(self, dim, hidden, p):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(p), nn.Linear(hidden, dim), nn.Dropout(p))

    def forward(self, x):
        return self.net(x)


class _VitAttn(nn.Module):
    def __init__(self, dim, heads, dim_head, p):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(p))

    def forward(self, x):
        b, n, _ = x.shape
        q, k, v = (t.view(b, n, self.heads, -1).transpose(1, 2) for t in self.to_qkv(x).chunk(3, dim=-1))
        att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * self.scale, dim=-1)
        out = torch.matmul(att, v).transpose(1, 2).reshape(b, n, -1)
        return self.to_out(out)


class _VitBlocks(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, p):
        super().__init__()
        self.layers = nn.ModuleList(
            nn.ModuleList([_PreNorm(dim, _VitAttn(dim, heads, dim_head, p)), _PreNorm(dim, _VitFF(dim, mlp_dim, p))])
            for _ in range(depth))

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return x


class _Patchify(nn.Module):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' without einops (index 0 of to_patch_embedding)."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, img):
        b, c, hh, ww = img.shape
        p = self.p
        x = img.view(b, c, hh // p, p, ww // p, p).permute(0, 2, 4, 3, 5, 1)
        return x.reshape(b, (hh // p) * (ww // p), p * p * c)


class MapViT(nn.Module):
    """vit_pytorch 0.33.2 ``ViT`` architecture (cls pooling, pre-norm blocks, dim_head 64)."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, dropout=0.0,
                 emb_dropout=0.0, channels=3, dim_head=64):
        super().__init__()
        n_patches = (image_size // patch_size) ** 2
        self.to_patch_embedding = nn.Sequential(_Patchify(patch_size), nn.Linear(channels * patch_size ** 2, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, n_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = _VitBlocks(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))

    def forward(self, img):
        x = self.to_patch_embedding(img)
        b, n, _ = x.shape
        x = torch.cat([self.cls_token.expand(b, -1, -1), x], dim=1) + self.pos_embedding[:, : n + 1]
        x = self.transformer(self.dropout(x))
        return self.mlp_head(x[:, 0])


# --------------------------------------------------------------------------- model graph
def build_modules(hp: dict, node_type: str = "VEHICLE") -> nn.ModuleDict:
    """Registry keys / ctor arguments as mgcvae.py:77-212 creates them."""
    tr = hp["transformer"]
    state_len = sum(len(v) for v in hp["state"][node_type].values())
    pred_len = sum(len(v) for v in hp["pred_state"][node_type].values())
    md = nn.ModuleDict()
    md[f"{node_type}/node_history_encoder"] = HistoryEncoder(
        ninp=state_len, nlayers=tr["nlayers"], nhead=tr["nhead"], in_dim=tr["in_dim"], fdim=tr["fdim"])
    if hp["lane_cnn_encoding"]:
        le = hp["lane_encoder"][node_type]
        md[f"{node_type}/lane_encoder"] = LaneConv(
            le["nlayers"], le["embedding_size"], le["map_channels"], le["output_size"], le["kernel_size"],
            le["strides"], hp["sample_num"])
        md[f"{node_type}/fusion/lane_hist"] = MlpStack(
            tr["in_dim"] + le["output_size"], tr["in_dim"], hp["fusion_hist_map_layer"], "regression")
        md[f"{node_type}/Lane/MLP_Softmax"] = MlpStack(
            le["output_size"] * 3, hp["max_lane_num"], hp["mlp_layer"], "classification")
    else:
        map_out = None
        if hp["map_vit_encoding"]:
            vp = hp["map_encoder"]["vit_param"]
            map_out = vp["output_size"]
            md[f"{node_type}/map_encoder"] = MapViT(
                image_size=vp["image_size"], patch_size=vp["patch_size"], num_classes=vp["output_size"],
                dim=vp["dim"], depth=vp["deep"], heads=vp["heads"], mlp_dim=vp["mlp_dim"],
                dropout=vp["dropout"], emb_dropout=vp["emb_dropout"])
        elif hp["map_cnn_encoding"]:
            cp = hp["map_encoder"]["cnn_param"]
            map_out = cp["output_size"]
            md[f"{node_type}/map_encoder"] = MapCNN(
                cp["map_channels"], cp["hidden_channels"], cp["output_size"], cp["masks"], cp["strides"],
                cp["patch_size"])
        if map_out is not None:
            md[f"{node_type}/fusion/hist_map"] = MlpStack(
                tr["output_size"] + map_out, tr["output_size"], hp["fusion_hist_map_layer"], "regression")
    md[f"{node_type}/decoder/MLP"] = MlpStack(tr["output_size"], pred_len, hp["mlp_layer"], "regression")
    md[f"{node_type}/decoder/transformer_decoder"] = TrajectoryDecoder(
        nlayers=tr["nlayers"], tgt_inp=pred_len, lane_inp=hp["lane_encoder"]["VEHICLE"]["output_size"],
        in_dim=tr["in_dim"], nhead=tr["nhead"], fdim=tr["fdim"], noutput=tr["output_size"])
    return md


class ReferenceFormModel:
    """Forward / loss graph of MultimodalGenerativeCVAE on top of ``build_modules``."""

    def __init__(self, hp: dict, modules: nn.ModuleDict | None = None, node_type: str = "VEHICLE"):
        self.hp = hp
        self.nt = node_type
        self.m = modules if modules is not None else build_modules(hp, node_type)

    def _mod(self, name):
        return self.m[f"{self.nt}/{name}"]

    # mgcvae.py:231-308
    def encode(self, x, x_st, lanes, maps, train: bool):
        pad = padding_mask(x_st)
        cmask = causal_mask(x_st.shape[-2], x_st.device)
        memory = self._mod("node_history_encoder")(x_st, cmask, pad)
        enc_map = enc_lane = None
        if self.hp["map_cnn_encoding"]:
            enc_map = self._mod("map_encoder")(maps * 2.0 - 1.0, train)
            enc_map = F.dropout(enc_map, self.hp["map_encoder"]["cnn_param"]["dropout"], training=train)
        elif self.hp["map_vit_encoding"]:
            enc_map = self._mod("map_encoder")(maps * 2.0 - 1.0)
        elif self.hp["lane_cnn_encoding"]:
            b = x.shape[0]
            ln, sn = self.hp["max_lane_num"], self.hp["sample_num"]
            emb = self.hp["lane_encoder"][self.nt]["embedding_size"]
            flat = lanes.view(b * ln, emb, sn)          # raw reinterpretation, not a transpose (:301-302)
            enc_lane = self._mod("lane_encoder")(flat).view(b, ln, -1)
        return memory, cmask, pad, x_st[:, -1], x[:, -1, 0:2], enc_lane, enc_map

    # mgcvae.py:375-449
    def ar_decode(self, memory, mem_mask, pad, n_s_t0, enc_lane, enc_map, ph, keep_h=None):
        dec = self._mod("decoder/transformer_decoder")
        head = self._mod("decoder/MLP")
        b = memory.shape[0]
        t_hist = memory.shape[-2]
        start = n_s_t0[:, 0:2].reshape(b, 1, -1)
        if self.hp["lane_cnn_encoding"]:
            fuse = self._mod("fusion/lane_hist")
            feat = enc_lane
            out = []
            for lane in range(enc_lane.shape[-2]):
                # the reference overwrites its own lane tensor here (:404): after lane 0 the
                # tensor is (B,T,32) with T identical rows, so index `lane` picks lane 0 again.
                feat = feat[:, [lane], :].repeat(1, t_hist, 1)
                mem = fuse(torch.cat([memory, feat], dim=-1))
                seq = start
                for step in range(ph):
                    h = dec(seq, mem, causal_mask(seq.shape[-2], seq.device), mem_mask, pad)
                    if step == ph - 1 and keep_h is not None:
                        keep_h.append(h)
                    seq = torch.cat([seq, (seq[:, -1, :] + head(h[:, -1, :])).view(b, 1, -1)], dim=-2)
                out.append(seq[:, 1:, :])
            return None, out
        fuse = self._mod("fusion/hist_map") if enc_map is not None else None
        seq = start
        for _ in range(ph):
            h = dec(seq, memory, causal_mask(seq.shape[-2], seq.device), mem_mask, pad)[:, [-1], :]
            if fuse is not None:
                h = fuse(torch.cat([h, enc_map.unsqueeze(-2)], dim=-1))
            seq = torch.cat([seq, head(h) + seq[:, [-1], :]], dim=-2)
        return seq[:, 1:, :], []

    # mgcvae.py:310-373  (teacher forced; only without --autoregressive)
    def tf_decode(self, memory, mem_mask, pad, labels_st, n_s_t0, enc_lane, enc_map, ph):
        dec = self._mod("decoder/transformer_decoder")
        head = self._mod("decoder/MLP")
        b = memory.shape[0]
        tgt = torch.cat([torch.zeros(b, 1, labels_st.shape[-1], device=memory.device), labels_st[:, :-1, :]], dim=-2)
        h = dec(tgt, memory, causal_mask(tgt.shape[-2], tgt.device), mem_mask, pad)
        if enc_map is not None:
            h = self._mod("fusion/hist_map")(torch.cat([h, enc_map.unsqueeze(-2).expand(-1, h.shape[1], -1)], dim=-1))
        return head(h).cumsum(dim=-2), None

    def _classify(self, keep_h):
        return self._mod("Lane/MLP_Softmax")(torch.cat(keep_h, dim=-1))

    # mgcvae.py:720-773
    def predict(self, x, x_st, lanes=None, maps=None, ph=12):
        memory, cm, pad, n0, pos, enc_lane, enc_map = self.encode(x, x_st, lanes, maps, train=False)
        keep = [] if self.hp["lane_cnn_encoding"] else None
        hist, lane = self.ar_decode(memory, cm, pad, n0, enc_lane, enc_map, ph, keep)
        if self.hp["lane_cnn_encoding"]:
            self._classify(keep)                        # computed and discarded, like the reference
            return None, torch.stack(lane, dim=1) * STD + pos.view(-1, 1, 1, 2)
        return hist * STD + pos.unsqueeze(1), None

    # mgcvae.py:559-627
    def train_loss(self, x, x_st, y, y_st, lanes=None, maps=None, ph=12, train=True):
        memory, cm, pad, n0, pos, enc_lane, enc_map = self.encode(x, x_st, lanes, maps, train=train)
        keep = [] if self.hp["lane_cnn_encoding"] else None
        if self.hp["autoregressive"]:
            hist, lane = self.ar_decode(memory, cm, pad, n0, enc_lane, enc_map, ph, keep)
        else:
            hist, lane = self.tf_decode(memory, cm, pad, y_st, n0, enc_lane, enc_map, ph)
        if self.hp["lane_cnn_encoding"]:
            self._classify(keep)
            per = torch.norm(torch.stack(lane, dim=1) - y_st.unsqueeze(1), p=2, dim=-1).sum(-1) / ph
            per = per.min(dim=-1)[0]
        else:
            per = torch.norm(hist - y_st, p=2, dim=-1).sum(-1) / ph
        return per.sum() / per.numel()

    # mgcvae.py:629-718
    def eval_loss(self, x, x_st, y, y_st, lanes=None, maps=None, ph=12):
        hist, lane = self.predict(x, x_st, lanes, maps, ph)
        if self.hp["lane_cnn_encoding"]:
            per = torch.norm(lane - y.unsqueeze(1), p=2, dim=-1).sum(-1) / ph
            per = per.min(dim=-1)[0]
        else:
            per = torch.norm(hist - y, p=2, dim=-1).sum(-1) / ph
        return per.sum() / per.numel()


def ade_fde(pred: torch.Tensor, gt: torch.Tensor):
    """evaluation/evaluation.py:10-19 batched: pred (B,ph,2) or (B,L,ph,2) [min over L, :153-156]."""
    if pred.dim() == 4:
        err = torch.norm(pred - gt.unsqueeze(1), dim=-1)
        return err.mean(-1).min(-1)[0], err[..., -1].min(-1)[0]
    err = torch.norm(pred - gt, dim=-1)
    return err.mean(-1), err[..., -1]
