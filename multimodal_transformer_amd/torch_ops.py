"""The Level-2 operators as PyTorch custom ops (namespace ``mmt``), registered with ``torch.library`` on top of the C-ABI calls
of ``ops.py`` (SURVEY.md 8b "native op surface"): visible to the dispatcher (``torch.ops.mmt.*``), with Meta kernels so that
shape propagation / fake-tensor tracing works, CUDA (= HIP) implementations only - there is no CPU kernel to dispatch to.

Packed weight descriptors (``ops.PackedEncoder`` ...) are Python objects holding device pointers; an op receives one as an
integer handle (``handle_of``), which stays valid while the descriptor is alive.

    memory, pad = torch.ops.mmt.encoder_fwd(x_st, handle_of(enc.packed()))
    pred_st     = torch.ops.mmt.decoder_ar_fwd(memory, pad, start_pos, None, None, handle_of(packed), 12)
"""
from __future__ import annotations

import weakref

import torch

from . import ops
from ._lib import MmtError

_HANDLES: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()


def handle_of(packed) -> int:
    """Integer handle of a packed descriptor (valid while the descriptor object is alive)."""
    h = id(packed)
    _HANDLES[h] = packed
    return h


def _get(handle, kind):
    obj = _HANDLES.get(int(handle))
    if obj is None or not isinstance(obj, kind):
        raise MmtError(f"stale or foreign handle {handle}: expected a live {kind.__name__}")
    return obj


_lib = torch.library.Library("mmt", "DEF")
_lib.define("encoder_fwd(Tensor x_st, int handle) -> (Tensor, Tensor)")
_lib.define("decoder_ar_fwd(Tensor memory, Tensor? pad_mask, Tensor start_pos, Tensor? map_feat, Tensor? mem_feat, int handle, int ph) -> Tensor")
_lib.define("decoder_tf_fwd(Tensor memory, Tensor? pad_mask, Tensor tgt, Tensor? mem_feat, int handle) -> Tensor")
_lib.define("cnn_map_fwd(Tensor maps, int handle, bool input_affine) -> Tensor")
_lib.define("vit_map_fwd(Tensor maps, int handle, bool input_affine) -> Tensor")
_lib.define("lane_enc_fwd(Tensor lanes, int handle) -> Tensor")
_lib.define("unstandardize(Tensor pred_st, Tensor node_pos) -> Tensor")
_lib.define("ade_fde(Tensor pred, Tensor gt) -> (Tensor, Tensor)")


def _encoder_fwd(x_st, handle):
    return ops.encoder_forward(_get(handle, ops.PackedEncoder), x_st, want_mask=True)


def _encoder_fwd_meta(x_st, handle):
    enc = _get(handle, ops.PackedEncoder)
    B, T, _ = x_st.shape
    return x_st.new_empty(B, T, enc.d_model), x_st.new_empty(B, T, dtype=torch.uint8)


def _decoder_ar_fwd(memory, pad_mask, start_pos, map_feat, mem_feat, handle, ph):
    return ops.decoder_forward(_get(handle, ops.PackedDecoder), memory, pad_mask, start_pos, ph, map_feat=map_feat, mem_feat=mem_feat)[0]


def _decoder_ar_fwd_meta(memory, pad_mask, start_pos, map_feat, mem_feat, handle, ph):
    dec = _get(handle, ops.PackedDecoder)
    return memory.new_empty(memory.shape[0], ph, dec.n_in)


def _decoder_tf_fwd(memory, pad_mask, tgt, mem_feat, handle):
    return ops.decoder_forward(_get(handle, ops.PackedDecoder), memory, pad_mask, tgt=tgt, mem_feat=mem_feat)[1]


def _decoder_tf_fwd_meta(memory, pad_mask, tgt, mem_feat, handle):
    dec = _get(handle, ops.PackedDecoder)
    return memory.new_empty(memory.shape[0], tgt.shape[1], dec.n_out)


def _cnn_map_fwd(maps, handle, input_affine):
    return ops.cnn_forward(_get(handle, ops.PackedCnn), maps, input_affine=input_affine)


def _vit_map_fwd(maps, handle, input_affine):
    return ops.vit_forward(_get(handle, ops.PackedVit), maps, input_affine=input_affine)


def _lane_enc_fwd(lanes, handle):
    return ops.lane_forward(_get(handle, ops.PackedLane), lanes)


def _n_out_meta(kind):
    def meta(x, handle, *flags):
        return x.new_empty(x.shape[0], _get(handle, kind).desc.n_out)
    return meta


def _unstandardize(pred_st, node_pos):
    return ops.finalize(pred_st, node_pos)[0]


def _ade_fde(pred, gt):
    _, ade, fde, _ = ops.finalize(pred, None, gt, metres=False, want_pred=False, want_metrics=True)
    return ade, fde


for _name, _fn, _meta in (
        ("encoder_fwd", _encoder_fwd, _encoder_fwd_meta),
        ("decoder_ar_fwd", _decoder_ar_fwd, _decoder_ar_fwd_meta),
        ("decoder_tf_fwd", _decoder_tf_fwd, _decoder_tf_fwd_meta),
        ("cnn_map_fwd", _cnn_map_fwd, _n_out_meta(ops.PackedCnn)),
        ("vit_map_fwd", _vit_map_fwd, _n_out_meta(ops.PackedVit)),
        ("lane_enc_fwd", _lane_enc_fwd, _n_out_meta(ops.PackedLane)),
        ("unstandardize", _unstandardize, lambda p, n: torch.empty_like(p)),
        ("ade_fde", _ade_fde, lambda p, g: (p.new_empty(p.shape[0]), p.new_empty(p.shape[0])))):
    _lib.impl(_name, _fn, "CUDA")
    _lib.impl(_name, _meta, "Meta")


# ---- what the module classes call: the same ops through the dispatcher, with this package's error for host tensors ------------
def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MmtError(f"tensor is on {t.device}; the HIP kernels need device memory (no CPU fallback)")


def encoder_fwd(packed, x_st):
    _dev(x_st)
    return torch.ops.mmt.encoder_fwd(x_st, handle_of(packed))


def decoder_ar_fwd(packed, memory, pad_mask, start_pos, ph, map_feat=None, mem_feat=None):
    _dev(memory, pad_mask, start_pos, map_feat, mem_feat)
    return torch.ops.mmt.decoder_ar_fwd(memory, pad_mask, start_pos, map_feat, mem_feat, handle_of(packed), int(ph))


def decoder_tf_fwd(packed, memory, pad_mask, tgt, mem_feat=None):
    _dev(memory, pad_mask, tgt, mem_feat)
    return torch.ops.mmt.decoder_tf_fwd(memory, pad_mask, tgt, mem_feat, handle_of(packed))


def cnn_map_fwd(packed, maps, input_affine):
    _dev(maps)
    return torch.ops.mmt.cnn_map_fwd(maps, handle_of(packed), bool(input_affine))


def vit_map_fwd(packed, maps, input_affine):
    _dev(maps)
    return torch.ops.mmt.vit_map_fwd(maps, handle_of(packed), bool(input_affine))


def lane_enc_fwd(packed, lanes):
    _dev(lanes)
    return torch.ops.mmt.lane_enc_fwd(lanes, handle_of(packed))


def unstandardize(pred_st, node_pos):
    _dev(pred_st, node_pos)
    return torch.ops.mmt.unstandardize(pred_st, node_pos)
