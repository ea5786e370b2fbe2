"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors generated from the
reference and against the CPU oracle on seeded inputs.

Tolerances (BASELINE.json north_star): per-output relative error <= 1e-4 on the kernel's native
(standardised) outputs, ADE/FDE within 1e-3 m.
"""
import numpy as np
import pytest
import torch

import helpers
from helpers import MODES, TINY_TR, batch_for, hp_for, load_golden, rel_err, set_weights
from multimodal_transformer_amd.config import make_hyperparams
from multimodal_transformer_amd.synthetic import make_batch
from multimodal_transformer_amd.weights import apply_recipe

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True, params=["auto", "persistent", "cluster_launches"])
def decoder_impl(request):
    """Every test runs on both decoder kernel families (include/mmt_amd.h: mmt_decoder_select): "auto" takes the
    cluster kernels wherever their geometry allows (d <= 256, <= 16 keys), "persistent" forces the one-launch kernel."""
    from multimodal_transformer_amd import ops
    ops.decoder_select(request.param)
    yield request.param
    ops.decoder_select("auto")


class _Env:
    robot_type = None


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _model(hp, dev):
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    reg = ModelRegistrar("/tmp", dev)
    return helpers.no_dropout(MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, dev, [])), reg


def _to(batch, dev):
    t = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    return t["x"], t["x_st"], t["y"], t["y_st"], t.get("lanes"), t.get("map")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tiny", [True, False])
def test_predict_matches_reference_golden(mode, tiny):
    dev = _dev()
    z, meta = load_golden(("tiny_" if tiny else "full_") + mode)
    hp, B, ph = hp_for(mode, tiny)
    model, reg = _model(hp, dev)
    set_weights(reg.model_dict, tiny, z)
    x, x_st, y, y_st, lanes, maps = _to(batch_for(mode, tiny), dev)
    hist, lane = model.predict(x, x_st, lanes, None, maps, ph)
    pred = (lane if mode == "lane_cnn" else hist).cpu().numpy()
    node = x[:, -1, 0:2].cpu().numpy()
    node = node[:, None, None, :] if mode == "lane_cnn" else node[:, None, :]
    assert np.isfinite(pred).all()
    assert rel_err((pred - node) / 80.0, (z["predict"] - node) / 80.0) <= TOL
    # encoder memory
    memory, pad = reg.model_dict["VEHICLE/node_history_encoder"].forward_with_mask(x_st)
    m = z["memory"].shape[0]
    assert rel_err(memory[:m].cpu().numpy(), z["memory"]) <= TOL
    assert int(pad.sum()) == 0
    # ADE / FDE through the device reduction
    from multimodal_transformer_amd.evaluation import batch_ade_fde
    ade, fde = batch_ade_fde(lane if mode == "lane_cnn" else hist, y)
    assert np.max(np.abs(ade.cpu().numpy() - z["ade"])) <= 1e-3
    assert np.max(np.abs(fde.cpu().numpy() - z["fde"])) <= 1e-3
    # losses (forward values)
    ev, _ = model.eval_loss("VEHICLE", x, x_st, lanes, None, None, y, y_st, maps, ph)
    assert abs(float(ev) - float(z["eval_loss"])) <= TOL * abs(float(z["eval_loss"]))
    tr, _ = model.train_loss(x, x_st, lanes, None, None, y, y_st, maps, ph)
    assert abs(float(tr) - float(z["train_loss"])) <= TOL * abs(float(z["train_loss"]))
    if "enc_map" in z.files:
        em = reg.model_dict["VEHICLE/map_encoder"].forward_raw(maps)
        assert rel_err(em[: z["enc_map"].shape[0]].cpu().numpy(), z["enc_map"]) <= TOL
    if "enc_lane" in z.files:
        el = reg.model_dict["VEHICLE/lane_encoder"](lanes.view(-1, 2, 10)).view(B, 3, -1)
        assert rel_err(el[: z["enc_lane"].shape[0]].cpu().numpy(), z["enc_lane"]) <= TOL


def test_teacher_forced_decoder_matches_reference_golden():
    dev = _dev()
    z, _ = load_golden("tiny_basic")
    hp, B, ph = hp_for("basic", True)
    model, reg = _model(hp, dev)
    set_weights(reg.model_dict, True, z)
    y_st = torch.from_numpy(z["in::y_st"]).to(dev)
    tgt = torch.cat([torch.zeros(B, 1, 2, device=dev), y_st[:, :-1]], dim=1).contiguous()
    memory = torch.from_numpy(z["memory"]).to(dev)
    pad = torch.zeros(B, memory.shape[1], dtype=torch.bool, device=dev)
    dec = reg.model_dict["VEHICLE/decoder/transformer_decoder"]
    h = dec(tgt, memory, None, None, pad)
    assert rel_err(h.cpu().numpy(), z["decoder_tf"]) <= TOL


def _oracle_predict(hp, reg, batch, ph, n=None):
    from oracle import restatement as rs     # checker only
    cpu_modules = {k: v for k, v in reg.model_dict.items()}
    w = {}
    alias = {"node_history_encoder": "enc", "decoder/transformer_decoder": "dec", "decoder/MLP": "head",
             "fusion/hist_map": "fuse_map", "fusion/lane_hist": "fuse_lane", "map_encoder": "map",
             "lane_encoder": "lane", "Lane/MLP_Softmax": "lane_cls"}
    for key, mod in cpu_modules.items():
        short = alias[key.split("/", 1)[1]]
        for pn, t in mod.state_dict().items():
            w[f"{short}.{pn}"] = t.detach().double().cpu()
    t = {k: torch.from_numpy(v[:n]).double() for k, v in batch.items()}
    return rs.predict(w, hp, t["x"], t["x_st"], t.get("lanes"), t.get("map"), ph).numpy()


@pytest.mark.parametrize("mode,B,hist,ph", [
    ("basic", 1, 8, 12), ("basic", 17, 8, 12), ("basic", 300, 8, 12), ("basic", 5, 4, 8), ("basic", 3, 19, 1),
    ("basic", 40, 15, 20), ("map_cnn", 9, 8, 12), ("lane_cnn", 7, 8, 12), ("lane_cnn", 33, 3, 5),
])
def test_predict_matches_fp64_oracle_on_ragged_shapes(mode, B, hist, ph):
    """Edge shapes the reference's own configs never hit: batch 1, batches that do not fill a workgroup
    tile, history lengths below/above one 16-row tile, horizon 1."""
    dev = _dev()
    hp = make_hyperparams(mode, history=hist, horizon=ph)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=3)
    batch = make_batch(B, hist, ph, seed=11, with_map=(mode == "map_cnn"), with_lanes=(mode == "lane_cnn"))
    x, x_st, y, y_st, lanes, maps = _to(batch, dev)
    h, lane = model.predict(x, x_st, lanes, None, maps, ph)
    got = (lane if mode == "lane_cnn" else h).cpu().numpy()
    n = min(B, 24)
    want = _oracle_predict(hp, reg, batch, ph, n)
    node = batch["x"][:n, -1, 0:2]
    node = node[:, None, None, :] if mode == "lane_cnn" else node[:, None, :]
    assert rel_err((got[:n] - node) / 80.0, (want - node) / 80.0) <= TOL


def test_lane_dedup_equals_three_decodes():
    dev = _dev()
    hp = make_hyperparams("lane_cnn", history=8, horizon=12, transformer=TINY_TR)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=5)
    x, x_st, y, y_st, lanes, maps = _to(make_batch(6, 8, 12, seed=2, with_lanes=True), dev)
    _, a = model.predict(x, x_st, lanes, None, None, 12)
    model.lane_dedup = False
    _, b = model.predict(x, x_st, lanes, None, None, 12)
    assert torch.equal(a, b)
    assert torch.equal(b[:, 0], b[:, 1]) and torch.equal(b[:, 0], b[:, 2])


def test_padded_history_is_finite_and_masks_are_reported():
    """Front-padded (NaN) histories are 'undefined in reference' (SURVEY §0.7); here: finite outputs and the
    padding mask = isnan(x_st[:,:,1])."""
    dev = _dev()
    hp = make_hyperparams("basic", history=8, horizon=12, transformer=TINY_TR)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=5)
    batch = make_batch(5, 8, 12, seed=2)
    batch["x_st"][1, :3] = np.nan
    batch["x_st"][4, :7] = np.nan
    x, x_st, *_ = _to(batch, dev)
    memory, pad = reg.model_dict["VEHICLE/node_history_encoder"].forward_with_mask(x_st)
    want = np.isnan(batch["x_st"][:, :, 1])
    assert np.array_equal(pad.cpu().numpy().astype(bool), want)
    hist, _ = model.predict(x, x_st, None, None, None, 12)
    assert torch.isfinite(hist).all()
    # un-padded agents are unaffected by their neighbours' padding
    clean = make_batch(5, 8, 12, seed=2)
    xc, xc_st, *_ = _to(clean, dev)
    hist_c, _ = model.predict(xc, xc_st, None, None, None, 12)
    assert torch.equal(hist[[0, 2, 3]], hist_c[[0, 2, 3]])


def test_batch_invariance_and_determinism():
    """Agents are independent: any split of the batch gives bit-identical rows (the multi-GPU sharding
    property), and repeated launches are bit-identical."""
    dev = _dev()
    hp = make_hyperparams("basic", history=8, horizon=12)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=0)
    x, x_st, *_ = _to(make_batch(128, 8, 12, seed=1), dev)
    full, _ = model.predict(x, x_st, None, None, None, 12)
    again, _ = model.predict(x, x_st, None, None, None, 12)
    assert torch.equal(full, again)
    parts = [model.predict(x[i:i + 16].contiguous(), x_st[i:i + 16].contiguous(), None, None, None, 12)[0]
             for i in range(0, 128, 16)]
    assert torch.equal(full, torch.cat(parts))
    odd = model.predict(x[:37].contiguous(), x_st[:37].contiguous(), None, None, None, 12)[0]
    assert torch.equal(full[:37], odd)
    # 129..256 agents: still 4-agent clusters, the weight-stationary pipeline in throughput mode (more clusters than stages)
    # (the encoder runs on other tiles there - one workgroup per agent tile instead of two: same arithmetic per row, other
    # summation order - so rows agree to rounding, not bit for bit; rows 128.. repeat rows 0..71 and must equal them exactly)
    twice = model.predict(torch.cat([x, x[:72]]), torch.cat([x_st, x_st[:72]]), None, None, None, 12)[0]
    assert rel_err(twice[:128].cpu().numpy(), full.cpu().numpy()) <= 1e-6
    assert torch.equal(twice[128:], twice[:72])


def _flat(out):
    """tensors of a predict() result (a tensor, None, or nested lists / tuples of them)"""
    if out is None:
        return []
    if torch.is_tensor(out):
        return [out]
    return [t for o in out for t in _flat(o)]


def test_fused_decoder_is_bit_identical_to_launch_per_phase():
    """The one-launch cluster decoder (in-kernel hand-offs between the workgroups of a cluster) runs the same arithmetic in the
    same order as the launch-per-phase form: outputs must be bit-identical, call after call, for full and ragged cluster
    counts, in all three modes, also with other work queued on a second stream (a stale read of a handed-off partial would
    show up as a difference)."""
    from multimodal_transformer_amd import ops
    dev = _dev()
    side = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    other = torch.cuda.Stream()
    for mode in ("basic", "map_cnn", "lane_cnn"):
        hp = make_hyperparams(mode, history=8, horizon=12)
        model, reg = _model(hp, dev)
        apply_recipe(reg.model_dict, seed=0)
        for B in (128, 37, 3) + ((256, 203) if mode == "basic" else ()):      # > 128: the pipeline in throughput mode
            x, x_st, _, _, lanes, maps = _to(make_batch(B, 8, 12, seed=B, with_map=(mode == "map_cnn"),
                                                        with_lanes=(mode == "lane_cnn")), dev)
            ops.decoder_select("cluster_launches")
            want = model.predict(x, x_st, lanes, None, maps, 12)
            for form in ("cluster_pipeline", "cluster_fused"):     # the weight-stationary pipeline, the every-phase fused kernel
                ops.decoder_select(form)
                for it in range(8):
                    if it % 2:
                        with torch.cuda.stream(other):
                            side.add_(1.0)                     # streams through the caches beside the decode
                    got = model.predict(x, x_st, lanes, None, maps, 12)
                    for w, g_ in zip(_flat(want), _flat(got)):
                        assert torch.isfinite(g_).all(), (form, mode, B, it)
                        assert torch.equal(w, g_), (form, mode, B, it)
            torch.cuda.synchronize()
            ops.decoder_status()


@pytest.mark.parametrize("nlayers,ph", [(1, 12), (2, 16), (2, 20), (3, 16), (8, 12)])
def test_pipeline_tag_generations_at_other_depths_and_horizons(nlayers, ph):
    """The tagged hand-off tells the generations of a slot apart with 2 bits (decoder_cluster.hip, "tag bits"; the host checks
    3L/8 + 1 <= 4 in pipe_applies).  Depths whose 3L phases per step are NOT a multiple of the 8 slots walk through every
    (slot, tag) alignment over a long horizon: 1, 2, 3 layers at the longest horizon the pipeline takes (16 keys), the deepest
    stack (8 layers: 4 generations in the window, the bound itself), and horizon 20 (beyond the 16-key kernels: the persistent
    kernel must take over, same results).  Bit-identical to the launch-per-phase kernels, and right against the fp64 oracle."""
    from multimodal_transformer_amd import ops
    dev = _dev()
    hp = make_hyperparams("basic", history=8, horizon=ph,
                          transformer={"nlayers": nlayers, "nhead": 8, "in_dim": 256, "fdim": 512, "output_size": 32})
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=nlayers)
    try:
        for B in (128, 21):
            batch = make_batch(B, 8, ph, seed=ph + B)
            x, x_st, *_ = _to(batch, dev)
            ops.decoder_select("cluster_launches" if ph <= 16 else "persistent")
            want = model.predict(x, x_st, None, None, None, ph)[0]
            ops.decoder_select("cluster_pipeline" if ph <= 16 else "auto")
            for it in range(4):
                got = model.predict(x, x_st, None, None, None, ph)[0]
                assert torch.equal(want, got), (nlayers, ph, B, it)
            assert ops.decoder_last_path()[0] == (5 if ph <= 16 else 1), ops.decoder_last_path()
            torch.cuda.synchronize()
            ops.decoder_status()
            n = min(B, 12)
            ref = _oracle_predict(hp, reg, batch, ph, n)
            node = batch["x"][:n, -1, 0:2][:, None, :]
            assert rel_err((got[:n].cpu().numpy() - node) / 80.0, (ref - node) / 80.0) <= TOL
    finally:
        ops.decoder_select("auto")


@pytest.mark.parametrize("form", ["cluster_pipeline", "cluster_fused"])
def test_fused_decoder_times_out_loudly(form):
    """Fault injection: one workgroup leaves the one-launch decoder early (mmt_debug_decoder_stall), so the workgroups that
    depend on it can never complete their wait.  Every wait is bounded: the launch ends (0.2 s), the final head launch turns the
    WHOLE call's outputs into NaN (never stale numbers) and raises the device's fault word; mmt_decoder_status reports it as
    MMT_ELAUNCH, clears it and keeps the one-launch forms off on this device, so the re-issued call (ops.checked does that)
    returns the right numbers.  Both one-launch forms: the weight-stationary pipeline and the every-phase fused kernel."""
    import time
    from multimodal_transformer_amd import _lib, ops
    from multimodal_transformer_amd._lib import MmtError
    dev = _dev()
    lib = _lib.load()
    hp = make_hyperparams("basic", history=8, horizon=12)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=0)
    x, x_st, *_ = _to(make_batch(64, 8, 12, seed=2), dev)
    ops.decoder_select(form)
    want, _ = model.predict(x, x_st, None, None, None, 12)
    ops.decoder_status()                                  # clean so far
    try:
        lib.mmt_debug_decoder_stall(1)
        t0 = time.perf_counter()
        bad, _ = model.predict(x, x_st, None, None, None, 12)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isnan(bad).all()
        assert 0.1 < dt < 5.0
        with pytest.raises(MmtError, match="gave up"):
            ops.decoder_status()
        ops.decoder_status()                              # reported once, then clear
        # the one-launch forms are now off on this device: the same call (stall still armed) runs launch-per-phase and is right
        again, _ = model.predict(x, x_st, None, None, None, 12)
        assert torch.equal(want, again)
        ops.decoder_status()
        # and the wrapper the callers use re-issues by itself
        lib.mmt_debug_decoder_stall(0)                    # re-arms the one-launch forms
        lib.mmt_debug_decoder_stall(1)
        with pytest.warns(RuntimeWarning, match="re-issuing"):
            healed, _ = ops.checked(lambda: model.predict(x, x_st, None, None, None, 12))
        assert torch.equal(want, healed)
    finally:
        lib.mmt_debug_decoder_stall(0)
    fresh, _ = model.predict(x, x_st, None, None, None, 12)
    assert torch.equal(want, fresh)
    ops.decoder_status()


def test_teacher_forced_decoder_full_width_agrees_across_kernel_families():
    """Trajectory_Decoder.forward (teacher-forced, Transformer_model.py:168-199) at the shipped width: the fused one-launch form ==
    the launch-per-phase form bit for bit, and both agree with the persistent kernel (different summation order) to 1e-5."""
    from multimodal_transformer_amd import ops
    dev = _dev()
    hp = make_hyperparams("basic", history=8, horizon=12)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=0)
    for B in (128, 37):
        b = make_batch(B, 8, 12, seed=3)
        x_st, y_st = torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev)
        memory, pad = reg.model_dict["VEHICLE/node_history_encoder"].forward_with_mask(x_st)
        tgt = torch.cat([torch.zeros(B, 1, 2, device=dev), y_st[:, :-1]], dim=1).contiguous()
        dec = reg.model_dict["VEHICLE/decoder/transformer_decoder"]
        out = {}
        for impl in ("cluster", "cluster_pipeline", "cluster_fused", "cluster_launches", "persistent"):
            ops.decoder_select(impl)
            out[impl] = dec(tgt, memory, None, None, pad.bool())
        assert torch.isfinite(out["cluster"]).all()
        assert torch.equal(out["cluster"], out["cluster_launches"])
        assert torch.equal(out["cluster_pipeline"], out["cluster_launches"])
        assert torch.equal(out["cluster_fused"], out["cluster_launches"])
        assert rel_err(out["cluster"].cpu().numpy(), out["persistent"].cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("tiny", [True, False])
def test_train_and_test_decoder_wrappers_match_oracle(tiny):
    """train_decoder / test_decoder (mgcvae.py:451-550): the dispatch wrappers with the reference's signatures.  Lane mode adds
    Lane/MLP_Softmax on the concatenated per-lane decoder states (softmax over the batch axis, MLP.py:18-21) - checked against the
    oracle's ar_decode(keep_h) + _classify (the positions are pinned by the predict goldens; the classification output has no
    fixture in the reference: parity unpinned beyond the stand-alone Mlp golden)."""
    from oracle import reference_form as rf     # checker only
    from multimodal_transformer_amd.model.model_utils import ModeKeys
    dev = _dev()
    z, _ = load_golden(("tiny_" if tiny else "full_") + "lane_cnn")
    hp, B, ph = hp_for("lane_cnn", tiny)
    model, reg = _model(hp, dev)
    set_weights(reg.model_dict, tiny, z)
    batch = batch_for("lane_cnn", tiny)
    x, x_st, y, y_st, lanes, maps = _to(batch, dev)
    memory, _, pad, n0, node_pos, enc_lane, enc_map = model.obtain_encoded_tensors(ModeKeys.EVAL, x, x_st, lanes, maps)
    hist, lane, idx = model.test_decoder(True, memory, None, pad, n0, enc_lane, enc_map, ph)
    hist2, lane2, idx2 = model.train_decoder(True, memory, None, pad, y_st, n0, enc_lane, enc_map, ph)
    assert hist is None and hist2 is None and len(lane) == 3 and idx.shape == (B, ph, 3)
    assert torch.equal(idx, idx2) and all(torch.equal(a, b) for a, b in zip(lane, lane2))
    assert model.class_input == []
    ref = rf.ReferenceFormModel(hp)
    set_weights(ref.m, tiny, z)
    ref.m.eval()
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    with torch.no_grad():
        mem, cm, rpad, rn0, _, rlane, _ = ref.encode(t["x"], t["x_st"], t["lanes"], None, train=False)
        keep = []
        _, want_lane = ref.ar_decode(mem, cm, rpad, rn0, rlane, None, ph, keep)
        want_idx = ref._classify(keep)
    for got, want in zip(lane, want_lane):
        assert rel_err(got.cpu().numpy(), want.numpy()) <= TOL
    assert rel_err(idx.cpu().numpy(), want_idx.numpy()) <= TOL
    assert np.allclose(idx.sum(dim=0).cpu().numpy(), 1.0, atol=1e-5)
    # Basic mode: no classification, the wrappers only dispatch
    hp_b, Bb, _ = hp_for("basic", True)
    mb, rb = _model(hp_b, dev)
    zb, _ = load_golden("tiny_basic")
    set_weights(rb.model_dict, True, zb)
    xb, xb_st, yb, yb_st, _, _ = _to(batch_for("basic", True), dev)
    memory, _, pad, n0, _, _, _ = mb.obtain_encoded_tensors(ModeKeys.EVAL, xb, xb_st, None, None)
    h_ar, l_ar, i_ar = mb.test_decoder(False, memory, None, pad, n0, None, None, ph)
    h_tf, _, i_tf = mb.train_decoder(False, memory, None, pad, yb_st, n0, None, None, ph)
    assert i_ar is None and i_tf is None and l_ar == []
    assert torch.equal(h_ar, mb.at_dec(memory, None, pad, n0, None, None, ph)[0])
    assert torch.equal(h_tf, mb.non_at_dec(memory, None, pad, yb_st, n0, None, None, ph)[0])


@pytest.mark.parametrize("mode", MODES)
def test_trajectron_batch_entry_points_match_reference_golden(mode):
    """Trajectron.{train_loss, eval_loss, predict_batch} on the reference's 10-tuple batch (trajectron.py:58-135): the tuple is
    unpacked, moved to the device and handed to the node model; lane batches arrive as a tuple of per-node (3,10,2) arrays
    (trajectron.py:62-65).  Values against the goldens of the real reference."""
    import types
    from multimodal_transformer_amd.model import ModelRegistrar
    from multimodal_transformer_amd.model.trajectron import Trajectron
    dev = _dev()
    z, _ = load_golden("tiny_" + mode)
    hp, B, ph = hp_for(mode, True)
    reg = ModelRegistrar("/tmp", dev)
    tr = Trajectron(reg, hp, None, dev)
    tr.set_environment(types.SimpleNamespace(NodeType=["VEHICLE"], get_edge_types=lambda: [], robot_type=None))
    helpers.no_dropout(tr.node_models_dict["VEHICLE"])          # the goldens are dropout-free
    set_weights(reg.model_dict, True, z)
    b = batch_for(mode, True)
    t = {k: torch.from_numpy(v) for k, v in b.items()}          # host tensors, like a DataLoader batch
    lanes = None
    if mode == "lane_cnn":
        lanes = [tuple(b["lanes"][i] for i in range(B)), None, None]
    batch = (torch.zeros(B, dtype=torch.long), t["x"], t["y"], t["x_st"], t["y_st"], None, None, None, lanes, t.get("map"))
    loss, reg_loss = tr.train_loss(batch, "VEHICLE")
    assert abs(float(loss) - float(z["train_loss"])) <= TOL * abs(float(z["train_loss"])) and float(reg_loss) == float(loss)
    ev, ev_reg = tr.eval_loss(batch, "VEHICLE")
    assert isinstance(ev, np.ndarray) and abs(float(ev) - float(z["eval_loss"])) <= TOL * abs(float(z["eval_loss"]))
    hist, lane = tr.predict_batch(batch, "VEHICLE", ph)
    pred = (lane if mode == "lane_cnn" else hist).cpu().numpy()
    node = b["x"][:, -1, 0:2]
    node = node[:, None, None, :] if mode == "lane_cnn" else node[:, None, :]
    assert rel_err((pred - node) / 80.0, (z["predict"] - node) / 80.0) <= TOL


def test_checkpoint_round_trip_after_the_kernels_have_run(tmp_path):
    """ModelRegistrar.save_models / load_models (model_registrar.py:58-75): ``model_registrar-<iter>.pt`` = the pickled ModuleDict.
    Saved AFTER a predict (packed device descriptors exist and must not be pickled), loaded into a fresh registrar: same keys, same
    tensors, same prediction bit for bit."""
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    dev = _dev()
    hp, B, ph = hp_for("map_cnn", True)
    z, _ = load_golden("tiny_map_cnn")
    model, reg = _model(hp, dev)
    reg.model_dir = str(tmp_path)
    set_weights(reg.model_dict, True, z)
    x, x_st, y, y_st, lanes, maps = _to(batch_for("map_cnn", True), dev)
    before = model.predict(x, x_st, lanes, None, maps, ph)[0]
    reg.save_models(7)
    assert (tmp_path / "model_registrar-7.pt").exists()
    reg2 = ModelRegistrar(str(tmp_path), dev)
    reg2.load_models(7)
    assert list(reg2.model_dict.keys()) == list(reg.model_dict.keys())
    for (ka, a), (kb, b) in zip(reg.model_dict.state_dict().items(), reg2.model_dict.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
    model2 = MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg2, hp, dev, [])
    assert torch.equal(model2.predict(x, x_st, lanes, None, maps, ph)[0], before)
    with pytest.raises(ValueError):
        reg2.get_model("VEHICLE/never_registered")


def test_large_batch_properties():
    """BASELINE config 5 size on one GPU (8192/8 = 1024 agents): finite, and rows equal the B=128 run."""
    dev = _dev()
    hp = make_hyperparams("basic", history=8, horizon=12)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=0)
    x, x_st, *_ = _to(make_batch(1024, 8, 12, seed=1), dev)
    big, _ = model.predict(x, x_st, None, None, None, 12)
    assert torch.isfinite(big).all()
    small, _ = model.predict(x[:128].contiguous(), x_st[:128].contiguous(), None, None, None, 12)
    # different tile shapes (32-row encoder tiles, 4 agents per decoder workgroup): same arithmetic per row
    assert rel_err(big[:128].cpu().numpy(), small.cpu().numpy()) <= 1e-6


@pytest.mark.parametrize("mode", ["lane_cnn", "map_cnn"])
def test_config5_shard_batch_1024(mode):
    """BASELINE config 5's per-GPU shard (Lane_encoding_CNN, 8192 agents over 8 GPUs = 1024) and the map configuration at the same
    batch: a 24-agent slice against the fp64 oracle, and every 128-agent slice against the same agents run as a batch of 128
    (other kernel family there: 4-agent clusters in one launch instead of 16-agent clusters per phase) to 1e-4."""
    dev = _dev()
    hp = make_hyperparams(mode, history=8, horizon=12)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=0)
    B = 1024
    batch = make_batch(B, 8, 12, seed=21, with_map=(mode == "map_cnn"), with_lanes=(mode == "lane_cnn"))
    x, x_st, y, y_st, lanes, maps = _to(batch, dev)
    h, lane = model.predict(x, x_st, lanes, None, maps, 12)
    got = (lane if mode == "lane_cnn" else h)
    assert torch.isfinite(got).all()
    node = batch["x"][:, -1, 0:2]
    node = node[:, None, None, :] if mode == "lane_cnn" else node[:, None, :]
    want = _oracle_predict(hp, reg, batch, 12, 24)
    assert rel_err((got[:24].cpu().numpy() - node[:24]) / 80.0, (want - node[:24]) / 80.0) <= TOL
    g = got.cpu().numpy()
    for i in range(0, B, 128):
        sl = slice(i, i + 128)
        hs, ls = model.predict(x[sl].contiguous(), x_st[sl].contiguous(), lanes[sl].contiguous() if lanes is not None else None, None,
                               maps[sl].contiguous() if maps is not None else None, 12)
        part = (ls if mode == "lane_cnn" else hs).cpu().numpy()
        assert rel_err((g[sl] - node[sl]) / 80.0, (part - node[sl]) / 80.0) <= TOL, i


def test_error_paths():
    from multimodal_transformer_amd._lib import MmtError
    dev = _dev()
    hp = make_hyperparams("basic", history=8, horizon=12, transformer=TINY_TR)
    model, reg = _model(hp, dev)
    x, x_st, *_ = _to(make_batch(4, 8, 12, seed=1), dev)
    with pytest.raises(MmtError):
        model.predict(x, x_st[:, :, :5].contiguous(), None, None, None, 12)       # wrong feature count
    with pytest.raises(MmtError):
        model.predict(x, x_st.double(), None, None, None, 12)                       # wrong dtype
    with pytest.raises(MmtError):
        model.predict(x, x_st[:0].contiguous(), None, None, None, 12)               # empty batch
    with pytest.raises(MmtError):
        model.predict(x, x_st, None, None, None, 64)                                # horizon beyond the kernel's cap


def test_vit_map_encoder_matches_cpu_restatement():
    """ViT parity is UNPINNED against the reference (vit_pytorch 0.33.2 is not available offline): the check is
    against oracle/reference_form.MapViT, a restatement of that package's published architecture."""
    from oracle import reference_form as rf
    dev = _dev()
    hp = make_hyperparams("map_vit", history=8, horizon=12)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=4)
    batch = make_batch(6, 8, 12, seed=5, with_map=True)
    x, x_st, y, y_st, lanes, maps = _to(batch, dev)
    vit = reg.model_dict["VEHICLE/map_encoder"]
    got = vit.forward_raw(maps).cpu().numpy()
    ref = rf.ReferenceFormModel(hp)
    apply_recipe(ref.m, seed=4)
    ref.m.eval()
    with torch.no_grad():
        want = ref.m["VEHICLE/map_encoder"](torch.from_numpy(batch["map"]) * 2.0 - 1.0).numpy()
        hist_want, _ = ref.predict(torch.from_numpy(batch["x"]), torch.from_numpy(batch["x_st"]), None,
                                   torch.from_numpy(batch["map"]), 12)
    assert rel_err(got, want) <= TOL
    # drop-in call form: forward(map*2-1)
    got2 = vit((maps * 2.0 - 1.0).contiguous()).cpu().numpy()
    assert rel_err(got2, want) <= TOL
    hist, _ = model.predict(x, x_st, None, None, maps, 12)
    node = batch["x"][:, -1, 0:2][:, None, :]
    assert rel_err((hist.cpu().numpy() - node) / 80.0, (hist_want.numpy() - node) / 80.0) <= TOL


def test_gmm2d_matches_reference_golden():
    """Transformer/model/components/gmm2d.py (dead code in the live path, named by the north star): every attribute and
    method against tensors produced by the reference class (tests/golden/gmm2d.npz; rsample with the normal noise and
    the component draws pinned)."""
    import os
    from helpers import GOLDEN
    from multimodal_transformer_amd.model.components import GMM2D
    from multimodal_transformer_amd._lib import MmtError
    dev = _dev()
    z = np.load(os.path.join(GOLDEN, "gmm2d.npz"))
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    g = GMM2D(t("log_pis"), t("mus"), t("log_sigmas"), t("corrs"))
    for attr, key in (("log_pis", "norm_log_pis"), ("sigmas", "sigmas"), ("one_minus_rho2", "one_minus_rho2"), ("L", "L")):
        assert rel_err(getattr(g, attr).cpu().numpy(), z[key]) <= 1e-5, attr
    assert rel_err(g.get_covariance_matrix().cpu().numpy(), z["cov"]) <= 1e-5
    assert rel_err(g.log_prob(t("value")).cpu().numpy(), z["log_prob"]) <= 1e-5
    assert rel_err(g.reg_los(t("value")).cpu().numpy(), z["reg_los"]) <= 1e-5
    got = g.rsample(torch.Size([5]), noise=t("rsample_z"), components=t("rsample_comp"))
    assert got.shape == z["rsample"].shape and rel_err(got.cpu().numpy(), z["rsample"]) <= 1e-5
    # the mesh has 0.01 spacing: the arg max may move by one cell where two cells tie to fp32 rounding
    mode = g.mode().cpu().numpy()
    assert mode.shape == z["mode"].shape and np.max(np.abs(mode - z["mode"])) <= 0.0101
    lp_ours = g.log_prob(torch.from_numpy(mode).to(dev)).cpu().numpy()
    lp_ref = g.log_prob(t("mode")).cpu().numpy()
    assert np.all(lp_ours >= lp_ref - 1e-5)
    g2 = GMM2D.from_log_pis_mus_cov_mats(t("log_pis"), t("mus"), t("cov"))
    assert rel_err(g2.log_sigmas.cpu().numpy(), z["from_cov_log_sigmas"]) <= 1e-5
    assert rel_err(g2.corrs.cpu().numpy(), z["from_cov_corrs"]) <= 1e-4
    g1 = GMM2D(t("log_pis")[..., :1].contiguous(), t("mus")[..., :2].contiguous(), t("log_sigmas")[..., :2].contiguous(),
               t("corrs")[..., :1].contiguous())
    assert np.array_equal(g1.mode().cpu().numpy(), z["mode_single"])
    # unpinned draw: inverse-CDF component selection follows the mixture weights
    torch.manual_seed(0)
    big = GMM2D(t("log_pis")[0, 0, 0].repeat(20000, 1), t("mus")[0, 0, 0].repeat(20000, 1),
                t("log_sigmas")[0, 0, 0].repeat(20000, 1) - 20.0, t("corrs")[0, 0, 0].repeat(20000, 1))
    s = big.rsample()
    mus = z["mus"][0, 0, 0].reshape(-1, 2)
    which = np.argmin(((s.cpu().numpy()[:, None, :] - mus[None]) ** 2).sum(-1), axis=1)
    freq = np.bincount(which, minlength=4) / 20000.0
    assert np.max(np.abs(freq - np.exp(z["norm_log_pis"][0, 0, 0]))) < 0.02
    with pytest.raises(MmtError):
        GMM2D(torch.zeros(2, 3), torch.zeros(2, 6), torch.zeros(2, 6), torch.zeros(2, 3))      # CPU tensors


def test_map_patch_extraction_matches_reference_golden():
    """GeometricMap.get_cropped_maps_from_scene_map_batch (environment/map.py:93-173) as one gather-rotate
    kernel, against patches produced by the reference's pad -> crop -> grid_sample chain (tests/golden/mapcrop.npz).
    Two rasters in one batch, border agents, rotations incl. none; one layer is per-pixel noise, so the tolerance is an
    absolute 2e-4 on values in [0,1] (the reference builds its sampling grid through fp32 matrix inverses)."""
    import os
    from helpers import GOLDEN
    from multimodal_transformer_amd.environment import GeometricMap
    from multimodal_transformer_amd._lib import MmtError
    dev = _dev()
    z = np.load(os.path.join(GOLDEN, "mapcrop.npz"))
    maps = [GeometricMap(z["raster0"], z["homography0"]), GeometricMap(z["raster1"], z["homography1"])]
    ml = [maps[w] for w in z["which"]]
    pts, rot = z["scene_pts"], torch.from_numpy(z["rotation"])
    for tag in ("ref", "small"):
        ps = [int(v) for v in z[f"{tag}::patch_size"]]
        got = GeometricMap.get_cropped_maps_from_scene_map_batch(ml, torch.from_numpy(pts), ps, rotation=rot, device=dev)
        want = z[f"{tag}::rotated"]
        assert tuple(got.shape) == want.shape
        assert np.max(np.abs(got.cpu().numpy() - want)) <= 2e-4, tag
        got0 = GeometricMap.get_cropped_maps_from_scene_map_batch(ml, pts, ps, rotation=None, device=dev)
        assert np.max(np.abs(got0.cpu().numpy() - z[f"{tag}::unrotated"])) <= 2e-4, tag
    # single-map convenience form and the hand-off to the CNN map encoder's input layout
    one = maps[0].get_cropped_maps(pts[:3], [50, 10, 50, 90], rotation=rot[:3], device=dev)
    assert np.max(np.abs(one.cpu().numpy() - z["ref::rotated"][:3])) <= 2e-4 and one.shape[1:] == (3, 100, 100)
    with pytest.raises(MmtError):
        maps[0].get_cropped_maps(pts[:3], [50, 10, 50, 90], device="cpu")


@pytest.mark.parametrize("mode", ["basic", "map_cnn", "lane_cnn"])
def test_graph_replayed_predict_is_bit_identical_to_eager(mode, decoder_impl):
    """MultimodalGenerativeCVAE.predict_graphed: the whole predict captured once per shape into a HIP graph --
    first call eager, second call captures, later calls replay -- equals the eager predict bit for bit, on the captured inputs
    and on new ones; a parameter update (version counter) drops the graph; mmt_decoder_status is clean after the replays."""
    from multimodal_transformer_amd import ops
    if decoder_impl == "persistent":
        pytest.skip("one family of the capture is enough: the persistent kernel is not the B=128 path")
    dev = _dev()
    hp, B, ph = hp_for(mode, False)
    model, reg = _model(hp, dev)
    apply_recipe(reg.model_dict, seed=0)
    batches = [_to(make_batch(B, 8, ph, seed=s, with_map=(mode == "map_cnn"), with_lanes=(mode == "lane_cnn")), dev) for s in (1, 2, 3)]
    pick = (lambda o: o[1]) if mode == "lane_cnn" else (lambda o: o[0])
    want = [pick(model.predict(b[0], b[1], b[4], None, b[5], ph)).clone() for b in batches]
    torch.cuda.synchronize()
    model.graph_predict = True
    got = []
    for i in (0, 0, 1, 2, 0):          # warm (eager), capture, replay x3
        b = batches[i]
        got.append((i, pick(model.predict(b[0], b[1], b[4], None, b[5], ph))))
    torch.cuda.synchronize()
    ops.decoder_status()
    ent = next(iter(model._predict_graphs.values()))
    assert ent["state"] == "ready" and len(model._predict_graphs) == 1
    for i, g in got:
        assert torch.equal(g, want[i]), (mode, i)
    assert got[2][1].data_ptr() != got[3][1].data_ptr()          # fresh tensors, as predict returns
    # an in-place parameter update is noticed: the stale graph is dropped, the next calls warm up and re-capture
    with torch.no_grad():
        reg.model_dict["VEHICLE/decoder/MLP"].layer_seq.out_mlp.bias.add_(0.125)
    b = batches[1]
    moved = pick(model.predict(b[0], b[1], b[4], None, b[5], ph))
    assert next(iter(model._predict_graphs.values()))["state"] == "warm"
    again = pick(model.predict(b[0], b[1], b[4], None, b[5], ph))
    replay = pick(model.predict(b[0], b[1], b[4], None, b[5], ph))
    torch.cuda.synchronize()
    assert not torch.equal(moved, want[1]) and torch.equal(moved, again) and torch.equal(moved, replay)
    model.graph_predict = False
    assert torch.equal(pick(model.predict(b[0], b[1], b[4], None, b[5], ph)), moved)
    ops.decoder_status()
