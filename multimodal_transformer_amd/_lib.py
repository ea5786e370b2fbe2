"""ctypes binding of libmmt_amd.so (declarations mirror include/mmt_amd.h one to one).

There is no fallback: if the shared library is missing or a kernel is asked to run on
anything but HIP device memory, the call raises.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C multimodal_transformer_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

MMT_MAX_LAYERS = 8
MMT_MAX_MLP_HIDDEN = 3
MMT_MAX_CONV = 4
MMT_DECODER_STATIC_SCRATCH_FLOATS = 11 << 20
MMT_DECODER_AUTO, MMT_DECODER_PERSISTENT, MMT_DECODER_CLUSTER, MMT_DECODER_CLUSTER_LAUNCHES = 0, 1, 2, 3
MMT_DECODER_CLUSTER_FUSED, MMT_DECODER_CLUSTER_PIPELINE = 4, 5

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMT_AMD_LIB", os.path.join(_HERE, "libmmt_amd.so"))   # override: diagnostic builds only

fptr = C.c_void_p


class Linear(C.Structure):
    _fields_ = [("w", fptr), ("b", fptr), ("n", C.c_int32), ("k", C.c_int32)]


class Norm(C.Structure):
    _fields_ = [("g", fptr), ("b", fptr)]


class EncLayer(C.Structure):
    _fields_ = [("qkv", Linear), ("out", Linear), ("ff1", Linear), ("ff2", Linear), ("ln1", Norm), ("ln2", Norm)]


class Encoder(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("nhead", C.c_int32), ("d_ff", C.c_int32), ("nlayers", C.c_int32),
                ("n_in", C.c_int32), ("input_fc", Linear), ("pe", fptr), ("layer", EncLayer * MMT_MAX_LAYERS)]


class DecLayer(C.Structure):
    _fields_ = [("qkv", Linear), ("out", Linear), ("cq", Linear), ("ckv", Linear), ("cout", Linear),
                ("ff1", Linear), ("ff2", Linear), ("ln1", Norm), ("ln2", Norm), ("ln3", Norm),
                ("qkv4", fptr), ("out4", fptr), ("cq4", fptr), ("cout4", fptr), ("ff14", fptr), ("ff24", fptr)]


class Mlp(C.Structure):
    _fields_ = [("n_hidden", C.c_int32), ("hidden", Linear * MMT_MAX_MLP_HIDDEN),
                ("hidden_ln", Norm * MMT_MAX_MLP_HIDDEN), ("out", Linear)]


class Decoder(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("nhead", C.c_int32), ("d_ff", C.c_int32), ("nlayers", C.c_int32),
                ("n_in", C.c_int32), ("n_out", C.c_int32), ("input_fc", Linear), ("output_fc", Linear), ("pe", fptr),
                ("layer", DecLayer * MMT_MAX_LAYERS), ("head", Mlp), ("has_map_fusion", C.c_int32),
                ("map_fusion", Mlp), ("has_mem_fusion", C.c_int32), ("mem_fusion", Mlp)]


class DecoderDropout(C.Structure):
    _fields_ = [("pe", fptr)] + [(n, fptr * MMT_MAX_LAYERS) for n in ("attn1", "drop1", "attn2", "drop2", "ff", "drop3")] + \
               [("pe_scale", C.c_float)] + [(n + "_scale", C.c_float * MMT_MAX_LAYERS)
                                            for n in ("attn1", "drop1", "attn2", "drop2", "ff", "drop3")]


class DecoderStashLayout(C.Structure):
    """mmt_decoder_stash_layout (include/mmt_amd.h): offsets, in floats, of the training stash the pipeline decode writes."""
    FIELDS = ("layer_floats", "tail", "total_floats", "x", "qkv", "a1", "xhat1", "x1", "cq", "a2", "xhat2", "x2", "xhat3", "f1",
              "rstd1", "rstd2", "rstd3", "p1", "p2", "t_x", "t_ho", "t_hu", "t_hxhat", "t_hrstd", "t_u", "ho_ld")
    _fields_ = [(n, C.c_size_t) for n in FIELDS]


class Cnn(C.Structure):
    _fields_ = [("nconv", C.c_int32), ("in_ch", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("out_ch", C.c_int32 * MMT_MAX_CONV), ("ksize", C.c_int32 * MMT_MAX_CONV),
                ("stride", C.c_int32 * MMT_MAX_CONV), ("conv_w", fptr * MMT_MAX_CONV), ("conv_b", fptr * MMT_MAX_CONV),
                ("fc_w", fptr), ("fc_b", fptr), ("n_out", C.c_int32), ("input_affine", C.c_int32),
                ("conv_wp", fptr * MMT_MAX_CONV)]


class Lane(C.Structure):
    _fields_ = [("in_ch", C.c_int32), ("out_ch", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32),
                ("n_samples", C.c_int32), ("n_out", C.c_int32), ("conv_w", fptr), ("conv_b", fptr), ("fc_w", fptr),
                ("fc_b", fptr)]


class VitBlock(C.Structure):
    _fields_ = [("ln1", Norm), ("ln2", Norm), ("qkv", Linear), ("out", Linear), ("ff1", Linear), ("ff2", Linear)]


class Vit(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("image", "patch", "channels", "dim", "depth", "heads", "dim_head", "mlp_dim",
                                         "n_out", "heads_per_group", "panel_k", "input_affine")] + \
               [("patch_fc", Linear), ("cls_token", fptr), ("pos_embedding", fptr),
                ("block", VitBlock * MMT_MAX_LAYERS), ("head_ln", Norm), ("head_fc", Linear)]


class GemmArgs(C.Structure):
    _fields_ = [("A", fptr), ("sam", C.c_int64), ("sak", C.c_int64), ("B", fptr), ("sbk", C.c_int64),
                ("sbn", C.c_int64), ("C", fptr), ("ldc", C.c_int64), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("bias", fptr), ("addend", fptr), ("ld_add", C.c_int64), ("mask", fptr),
                ("ld_mask", C.c_int64), ("relu", C.c_int32), ("accumulate", C.c_int32), ("ln_xhat", fptr),
                ("ln_rstd", fptr), ("ln_g", fptr), ("ln_dx", fptr), ("ld_ln", C.c_int64), ("splitk", C.c_int32),
                ("pad_", C.c_int32), ("part_stride", C.c_int64), ("asum", fptr),
                ("drop", fptr), ("ld_drop", C.c_int64), ("drop_scale", C.c_float), ("drop_after_add", C.c_int32),
                ("mask_scale", C.c_float), ("pad2_", C.c_int32), ("ln_drop", fptr), ("ld_ln_drop", C.c_int64),
                ("ln_dxm", fptr), ("ln_drop_scale", C.c_float), ("pad3_", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [("q", fptr), ("k", fptr), ("v", fptr), ("o", fptr), ("probs", fptr), ("pad", fptr)] + \
               [(n, C.c_int64) for n in ("q_bs", "q_rs", "k_bs", "k_rs", "v_bs", "v_rs", "o_bs", "o_rs", "p_hs", "p_rs")] + \
               [(n, C.c_int32) for n in ("B", "H", "Lq", "Lk", "dh", "causal", "causal_offset")] + [("scale", C.c_float)] + \
               [("pdrop", fptr), ("pdrop_scale", C.c_float), ("pad_", C.c_int32)]


class PackJob(C.Structure):
    _fields_ = [("src", fptr), ("dst", fptr), ("total", C.c_int64)] + \
               [(n, C.c_int32) for n in ("kind", "n", "k", "ldw", "col0", "nparts", "part_rows", "slice_rows", "nslices", "tr")]


class SweepLayer(C.Structure):
    _fields_ = [("w1", fptr * 3), ("w2", fptr * 3), ("w2k", fptr), ("w2v", fptr), ("gamma", fptr * 3), ("xhat", fptr * 3),
                ("rstd", fptr * 3), ("gdy", fptr * 3), ("gdx", fptr * 3), ("f1", fptr), ("gf1", fptr), ("cq", fptr), ("mkv", fptr),
                ("p2", fptr), ("gcq", fptr), ("gmkv", fptr), ("qkv", fptr), ("p1", fptr), ("gqkv", fptr),
                ("k_ln", fptr * 3), ("k_ff", fptr), ("k_att", fptr * 2), ("s_ln", C.c_float * 3), ("s_ff", C.c_float),
                ("s_att", C.c_float * 2)]


class SweepPipe(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "T", "ph", "d", "H", "nlayers", "d_ff", "n_out", "n_pos", "pad_")] + \
               [("layer", SweepLayer * MMT_MAX_LAYERS)] + \
               [(n, fptr) for n in ("G", "gx0", "w_in", "w_hout", "h_gamma", "w_h0", "w_ofc", "h_hu", "h_xhat", "h_rstd", "h_ghum",
                                    "h_ghz", "h_gho", "poison", "k_pe")] + \
               [("s_pe", C.c_float), ("pad2_", C.c_float), ("w_fuse", fptr), ("h_gu", fptr), ("gho_ld", C.c_int32), ("map_dim", C.c_int32)]


class LnGradJob(C.Structure):
    _fields_ = [("dy", fptr), ("xhat", fptr), ("gw", fptr), ("gb", fptr), ("ld", C.c_int64), ("part_stride", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("splitk", C.c_int32), ("pad_", C.c_int32)]


class AdamwTensor(C.Structure):
    _fields_ = [("param", fptr), ("grad", fptr), ("exp_avg", fptr), ("exp_avg_sq", fptr), ("numel", C.c_int64),
                ("lr_scale", C.c_float), ("pad_", C.c_int32)]


# symbol -> (restype, argtypes); tests check that the library exports exactly these
SIGNATURES = {
    "mmt_last_error": (C.c_char_p, []),
    "mmt_version": (C.c_int, []),
    "mmt_packed_linear_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "mmt_pack_linear": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr, C.c_void_p]),
    "mmt_batch_gather": (C.c_int, [fptr, fptr, fptr, fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                   C.c_int, C.c_double, fptr, fptr, fptr, fptr, fptr, C.c_void_p]),
    "mmt_lane_gather": (C.c_int, [fptr, fptr, fptr, fptr, fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                  fptr, fptr, fptr, C.c_void_p]),
    "mmt_pack_jobs": (C.c_int, [fptr, C.c_int, C.c_void_p]),
    "mmt_lin4_first_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mmt_pack_lin4_first": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fptr, C.c_void_p]),
    "mmt_lin4_second_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mmt_pack_lin4_second": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr, C.c_void_p]),
    "mmt_encoder_forward": (C.c_int, [C.POINTER(Encoder), fptr, C.c_int, C.c_int, fptr, fptr, C.c_void_p]),
    "mmt_encoder_workspace_floats": (C.c_size_t, [C.POINTER(Encoder), C.c_int, C.c_int]),
    "mmt_encoder_forward_ws": (C.c_int, [C.POINTER(Encoder), fptr, C.c_int, C.c_int, fptr, fptr, fptr, C.c_void_p]),
    "mmt_decoder_workspace_floats": (C.c_size_t, [C.POINTER(Decoder), C.c_int, C.c_int, C.c_int]),
    "mmt_decoder_select": (C.c_int, [C.c_int]),
    "mmt_decoder_forward": (C.c_int, [C.POINTER(Decoder), fptr, fptr, fptr, fptr, C.c_int, fptr, C.c_int, fptr,
                                      C.c_int, C.c_int, C.c_int, fptr, fptr, fptr, C.c_void_p]),
    "mmt_decoder_forward_static": (C.c_int, [C.POINTER(Decoder), fptr, fptr, fptr, fptr, C.c_int, fptr, C.c_int, fptr,
                                             C.c_int, C.c_int, C.c_int, fptr, fptr, C.c_void_p]),
    "mmt_decoder_dropout_supported": (C.c_int, [C.POINTER(Decoder), C.c_int, C.c_int, C.c_int]),
    "mmt_decoder_forward_dropout": (C.c_int, [C.POINTER(Decoder), fptr, fptr, fptr, fptr, C.c_int, fptr, C.c_int, C.c_int, C.c_int,
                                              C.c_int, fptr, fptr, C.POINTER(DecoderDropout), C.c_void_p]),
    "mmt_decoder_stash_supported": (C.c_int, [C.POINTER(Decoder), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mmt_decoder_stash_layout_of": (C.c_int, [C.POINTER(Decoder), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(DecoderStashLayout)]),
    "mmt_decoder_forward_stash": (C.c_int, [C.POINTER(Decoder), fptr, fptr, fptr, fptr, C.c_int, fptr, C.c_int, C.c_int, C.c_int, C.c_int,
                                            fptr, fptr, C.POINTER(DecoderDropout), fptr, C.c_size_t, C.c_void_p]),
    "mmt_sweep_pipe_supported": (C.c_int, [C.POINTER(SweepPipe)]),
    "mmt_sweep_pipe_workspace_floats": (C.c_size_t, [C.POINTER(SweepPipe)]),
    "mmt_sweep_pipe_run": (C.c_int, [C.POINTER(SweepPipe), fptr, C.c_void_p]),
    "mmt_decoder_status": (C.c_int, [C.c_void_p]),
    "mmt_decoder_last_path": (C.c_int, []),
    "mmt_debug_decoder_stall": (C.c_int, [C.c_int]),
    "mmt_packed_conv_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mmt_pack_conv": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, fptr, C.c_void_p]),
    "mmt_cnn_workspace_floats": (C.c_size_t, [C.POINTER(Cnn), C.c_int]),
    "mmt_cnn_forward": (C.c_int, [C.POINTER(Cnn), fptr, C.c_int, fptr, fptr, C.c_void_p]),
    "mmt_vit_forward": (C.c_int, [C.POINTER(Vit), fptr, C.c_int, fptr, C.c_void_p]),
    "mmt_lane_forward": (C.c_int, [C.POINTER(Lane), fptr, C.c_int, fptr, C.c_void_p]),
    "mmt_gemm": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "mmt_gemm_attention_bwd": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(AttnArgs), fptr, C.c_long, fptr, fptr, C.c_long, C.c_long,
                                         C.c_void_p]),
    "mmt_im2col": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_float,
                             C.c_float, C.c_float, fptr, C.c_long, C.c_void_p]),
    "mmt_col2im": (C.c_int, [fptr, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fptr, C.c_float, fptr,
                             C.c_long, C.c_void_p]),
    "mmt_ln_param_grads": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_long, fptr, fptr, C.c_int, C.c_long, C.c_void_p]),
    "mmt_gemm_group_plan": (C.c_int, [C.POINTER(GemmArgs), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "mmt_gemm_group": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mmt_ln_grad_group_plan": (C.c_int, [C.POINTER(LnGradJob), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_int)]),
    "mmt_ln_grad_group": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mmt_colsum": (C.c_int, [fptr, fptr, fptr, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, fptr, C.c_int, C.c_void_p]),
    "mmt_layernorm_fwd": (C.c_int, [fptr, fptr, C.c_int, C.c_int, fptr, fptr, C.c_int, fptr, fptr, fptr, C.c_void_p]),
    "mmt_layernorm_bwd": (C.c_int, [fptr, fptr, fptr, fptr, fptr, C.c_int, C.c_int, fptr, C.c_int, fptr, fptr, C.c_float, fptr,
                                    C.c_void_p]),
    "mmt_dropout_mask": (C.c_int, [fptr, C.c_long, C.c_float, fptr, C.c_uint32, C.c_void_p]),
    "mmt_rng_advance": (C.c_int, [fptr, C.c_void_p]),
    "mmt_dropout_apply": (C.c_int, [fptr, fptr, C.c_float, fptr, C.c_long, C.c_void_p]),
    "mmt_attention_fwd": (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    "mmt_attention_bwd": (C.c_int, [C.POINTER(AttnArgs), fptr, C.c_long, C.c_long, fptr, C.c_long, C.c_long, fptr, fptr,
                                    C.c_long, C.c_long, C.c_void_p]),
    "mmt_l2_loss_bwd": (C.c_int, [fptr, fptr, C.c_int, C.c_int, fptr, fptr, C.c_void_p]),
    "mmt_adamw_multi": (C.c_int, [fptr, C.c_int, fptr, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  fptr, C.c_void_p]),
    "mmt_history_prepare": (C.c_int, [fptr, C.c_int, C.c_int, fptr, fptr, C.c_void_p]),
    "mmt_lr_step": (C.c_int, [fptr, C.c_float, fptr, fptr, C.c_void_p]),
    "mmt_softmax_axis": (C.c_int, [fptr, C.c_long, C.c_int, C.c_long, fptr, C.c_void_p]),
    "mmt_cumsum_axis": (C.c_int, [fptr, C.c_long, C.c_int, C.c_long, fptr, C.c_void_p]),
    "mmt_l2_norm": (C.c_int, [fptr, fptr, C.c_long, C.c_int, C.c_long, C.c_long, fptr, C.c_void_p]),
    "mmt_add_inplace": (C.c_int, [fptr, fptr, C.c_long, C.c_void_p]),
    "mmt_patchify": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, fptr, C.c_long, C.c_void_p]),
    "mmt_gelu_fwd": (C.c_int, [fptr, fptr, C.c_long, C.c_void_p]),
    "mmt_gelu_bwd": (C.c_int, [fptr, fptr, fptr, C.c_long, C.c_void_p]),
    "mmt_copy3": (C.c_int, [fptr, C.c_long, C.c_long, fptr, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mmt_gmm2d_prepare": (C.c_int, [fptr, fptr, fptr, C.c_long, C.c_int, fptr, fptr, fptr, fptr, fptr, C.c_void_p]),
    "mmt_gmm2d_log_prob": (C.c_int, [fptr, fptr, fptr, fptr, fptr, fptr, fptr, C.c_long, C.c_int, fptr, fptr, C.c_void_p]),
    "mmt_gmm2d_rsample": (C.c_int, [fptr, fptr, fptr, fptr, fptr, fptr, C.c_long, C.c_long, C.c_int, fptr, fptr, C.c_void_p]),
    "mmt_gmm2d_mode": (C.c_int, [fptr, fptr, fptr, fptr, fptr, fptr, C.c_long, C.c_int, fptr, C.c_void_p]),
    "mmt_gmm2d_from_cov": (C.c_int, [fptr, C.c_long, fptr, fptr, C.c_void_p]),
    "mmt_map_crop": (C.c_int, [fptr, fptr, C.c_int, fptr, fptr, fptr, C.POINTER(C.c_int32), C.c_int, fptr, C.c_void_p]),
    "mmt_finalize": (C.c_int, [fptr, fptr, fptr, C.c_int, C.c_int, C.c_int, fptr, fptr, fptr, fptr, C.c_void_p]),
}

_lib = None


class MmtError(RuntimeError):
    pass


def load():
    """Load libmmt_amd.so; raises (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmtError(
            f"{LIB_PATH} is not built: run `make -C multimodal_transformer_amd/csrc` (or __graft_entry__.build()). "
            "This package has no CPU/PyTorch fallback for its kernels.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mmt_last_error()
        raise MmtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
