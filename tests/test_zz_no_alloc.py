"""mmt_decoder_forward_static (include/mmt_amd.h): the decode for callers that cannot allocate -- its workspace is a file-scope
device array inside the library.  Called through the C ABI and compared bit for bit with mmt_decoder_forward (caller workspace)
on two calls in a row, the second with a smaller batch (the array keeps the first call's contents), for the one-launch,
launch-per-phase and persistent kernel families; sizes beyond the array are rejected.

Collected last on purpose (file name): the round's end runs `pytest -x`.
"""
import ctypes as C

import pytest
import torch

from multimodal_transformer_amd.config import make_hyperparams
from multimodal_transformer_amd.synthetic import make_batch
from multimodal_transformer_amd.weights import apply_recipe

pytestmark = pytest.mark.gpu


class _Env:
    robot_type = None


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _decode(lib, desc, memory, pad, start, ph, static, ws=None):
    B, T, _ = memory.shape
    pred = torch.full((B, ph, 2), float("nan"), dtype=torch.float32, device=memory.device)
    h = torch.full((B, ph, desc.n_out), float("nan"), dtype=torch.float32, device=memory.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if static:
        rc = lib.mmt_decoder_forward_static(C.byref(desc), _ptr(memory), _ptr(pad), _ptr(start), None, 0, None, 0, None, B, T, ph,
                                            _ptr(pred), _ptr(h), st)
    else:
        rc = lib.mmt_decoder_forward(C.byref(desc), _ptr(memory), _ptr(pad), _ptr(start), None, 0, None, 0, None, B, T, ph,
                                     _ptr(pred), _ptr(h), _ptr(ws), st)
    return rc, pred, h


@pytest.mark.parametrize("impl", ["auto", "cluster_launches", "persistent"])
def test_static_scratch_decode_equals_workspace_decode(impl):
    from multimodal_transformer_amd import _lib, ops
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    hp = make_hyperparams("basic", history=8, horizon=12)
    reg = ModelRegistrar("/tmp", dev)
    MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, dev, [])
    apply_recipe(reg.model_dict, seed=0)
    enc = reg.model_dict["VEHICLE/node_history_encoder"]
    dec = reg.model_dict["VEHICLE/decoder/transformer_decoder"]
    packed = dec.packed(head=reg.model_dict["VEHICLE/decoder/MLP"])
    ops.decoder_select(impl)
    try:
        for B in (128, 37):                       # two calls in a row, the second smaller
            b = make_batch(B, 8, 12, seed=B)
            x_st = torch.from_numpy(b["x_st"]).to(dev)
            memory, pad = enc.forward_with_mask(x_st)
            start = x_st[:, -1, 0:2].contiguous()
            nws = lib.mmt_decoder_workspace_floats(C.byref(packed.desc), B, 9, 12)
            assert 0 < nws <= _lib.MMT_DECODER_STATIC_SCRATCH_FLOATS
            ws = torch.empty(nws, dtype=torch.float32, device=dev)
            rc, want_p, want_h = _decode(lib, packed.desc, memory, pad, start, 12, static=False, ws=ws)
            assert rc == 0, lib.mmt_last_error()
            rc, got_p, got_h = _decode(lib, packed.desc, memory, pad, start, 12, static=True)
            assert rc == 0, lib.mmt_last_error()
            torch.cuda.synchronize()
            assert torch.isfinite(got_p).all() and torch.isfinite(got_h).all()
            assert torch.equal(want_p, got_p) and torch.equal(want_h, got_h)       # bit-exact (same kernels, same order)
        ops.decoder_status()
        # beyond the static array: rejected loudly, nothing launched
        big = 4096
        mem_big = torch.zeros(big, 9, 256, dtype=torch.float32, device=dev)
        start_big = torch.zeros(big, 2, dtype=torch.float32, device=dev)
        assert lib.mmt_decoder_workspace_floats(C.byref(packed.desc), big, 9, 12) > _lib.MMT_DECODER_STATIC_SCRATCH_FLOATS
        rc, p_big, _ = _decode(lib, packed.desc, mem_big, None, start_big, 12, static=True)
        assert rc == -1 and b"static buffer holds" in lib.mmt_last_error()
        assert torch.isnan(p_big).all()
    finally:
        ops.decoder_select("auto")


def test_static_scratch_decode_replays_from_a_hip_graph():
    """The advertised caller: after one eager call (the per-device set-up queries are not stream operations, include/mmt_amd.h),
    a decode captured into a HIP graph replays bit for bit -- also on new inputs written into the captured buffers."""
    from multimodal_transformer_amd import _lib, ops
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    dev = torch.device("cuda:0")
    lib = _lib.load()
    hp = make_hyperparams("basic", history=8, horizon=12)
    reg = ModelRegistrar("/tmp", dev)
    MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, dev, [])
    apply_recipe(reg.model_dict, seed=0)
    enc = reg.model_dict["VEHICLE/node_history_encoder"]
    dec = reg.model_dict["VEHICLE/decoder/transformer_decoder"]
    packed = dec.packed(head=reg.model_dict["VEHICLE/decoder/MLP"])
    B = 128
    xs = [torch.from_numpy(make_batch(B, 8, 12, seed=s)["x_st"]).to(dev) for s in (5, 6)]
    memory, pad = enc.forward_with_mask(xs[0])
    memory, pad = memory.clone(), pad.clone()
    start = xs[0][:, -1, 0:2].contiguous()
    rc, want0, _ = _decode(lib, packed.desc, memory, pad, start, 12, static=True)         # eager first call: set-up happens here
    assert rc == 0, lib.mmt_last_error()
    torch.cuda.synchronize()
    pred = torch.empty(B, 12, 2, dtype=torch.float32, device=dev)
    h = torch.empty(B, 12, packed.desc.n_out, dtype=torch.float32, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.mmt_decoder_forward_static(C.byref(packed.desc), _ptr(memory), _ptr(pad), _ptr(start), None, 0, None, 0, None,
                                            B, 9, 12, _ptr(pred), _ptr(h), st)
    assert rc == 0, lib.mmt_last_error()
    pred.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(pred, want0)
    m1, p1 = enc.forward_with_mask(xs[1])
    memory.copy_(m1), pad.copy_(p1), start.copy_(xs[1][:, -1, 0:2])
    g.replay()
    torch.cuda.synchronize()
    got1 = pred.clone()
    rc, want1, _ = _decode(lib, packed.desc, memory, pad, start, 12, static=True)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(got1, want1) and not torch.equal(want1, want0)
    ops.decoder_status()
