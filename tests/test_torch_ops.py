"""The Level-2 operators as torch.library custom ops (multimodal_transformer_amd/torch_ops.py)."""
import time

import pytest
import torch

from multimodal_transformer_amd.config import make_hyperparams
from multimodal_transformer_amd.synthetic import make_batch
from multimodal_transformer_amd.weights import apply_recipe


def test_ops_are_registered_and_have_no_cpu_kernel():
    from multimodal_transformer_amd import torch_ops  # noqa: F401
    for name in ("encoder_fwd", "decoder_ar_fwd", "decoder_tf_fwd", "cnn_map_fwd", "vit_map_fwd", "lane_enc_fwd", "unstandardize", "ade_fde"):
        assert hasattr(torch.ops.mmt, name)
    with pytest.raises(NotImplementedError):           # the product path has no CPU fallback, the dispatcher says so
        torch.ops.mmt.unstandardize(torch.zeros(2, 3, 2), torch.zeros(2, 2))
    with torch.device("meta"):                          # shape propagation without a kernel
        out = torch.ops.mmt.unstandardize(torch.empty(5, 12, 2), torch.empty(5, 2))
        ade, fde = torch.ops.mmt.ade_fde(torch.empty(5, 12, 2), torch.empty(5, 12, 2))
    assert out.shape == (5, 12, 2) and ade.shape == (5,) and fde.shape == (5,)


class _Env:
    robot_type = None


@pytest.mark.gpu
def test_custom_ops_equal_the_direct_calls():
    from multimodal_transformer_amd import ops, torch_ops
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    dev = torch.device("cuda:0")
    hp = make_hyperparams("map_cnn", history=8, horizon=12)
    reg = ModelRegistrar("/tmp", dev)
    model = MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, dev, [])
    apply_recipe(reg.model_dict, seed=0)
    b = make_batch(64, 8, 12, seed=4, with_map=True)
    t = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    want, _ = model.predict(t["x"], t["x_st"], None, None, t["map"], 12)
    enc = reg.model_dict["VEHICLE/node_history_encoder"].packed()
    cnn = reg.model_dict["VEHICLE/map_encoder"]._pk(t["map"])
    dec = reg.model_dict["VEHICLE/decoder/transformer_decoder"].packed(
        head=reg.model_dict["VEHICLE/decoder/MLP"], map_fusion=reg.model_dict["VEHICLE/fusion/hist_map"])
    h_enc, h_cnn, h_dec = torch_ops.handle_of(enc), torch_ops.handle_of(cnn), torch_ops.handle_of(dec)

    def via_ops():
        memory, pad = torch.ops.mmt.encoder_fwd(t["x_st"], h_enc)
        feat = torch.ops.mmt.cnn_map_fwd(t["map"], h_cnn, True)
        pred = torch.ops.mmt.decoder_ar_fwd(memory, pad, t["x_st"][:, -1, 0:2].contiguous(), feat, None, h_dec, 12)
        return torch.ops.mmt.unstandardize(pred, t["x"][:, -1, 0:2].contiguous())

    got = via_ops()
    assert torch.equal(want, got)
    ade, fde = torch.ops.mmt.ade_fde(got, t["y"])
    from multimodal_transformer_amd.evaluation import batch_ade_fde
    a2, f2 = batch_ade_fde(got, t["y"])
    assert torch.equal(ade, a2) and torch.equal(fde, f2)
    # dispatcher overhead per op (host side), for DESIGN.md
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        ops.finalize(got, t["x"][:, -1, 0:2].contiguous())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        torch.ops.mmt.unstandardize(got, t["x"][:, -1, 0:2].contiguous())
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"\nper call: direct {1e6 * (t1 - t0) / n:.1f} us, torch.ops.mmt {1e6 * (t2 - t1) / n:.1f} us")
    with pytest.raises(Exception):
        torch.ops.mmt.encoder_fwd(t["x_st"], 12345)           # stale handle
