"""Dropout in the training step (Transformer_model.py:28-30,77-79,126-129; mgcvae.py:291-293).

CPU tier: the mask-injecting oracle (oracle/dropout_form.py) is pinned against the stock-module reference form, the Philox
restatement against the published known-answer vectors, and the engine's host orchestration (which mask goes where, forward
and backward) against the oracle's autograd on the torch stand-in kernels.
GPU tier (-m gpu): the same comparison on the HIP kernels, the mask kernel bit for bit against the Philox restatement, the
keep rate, fresh masks per step, and ``obtain_encoded_tensors(TRAIN)`` with a map.
"""
import numpy as np
import pytest
import torch

import fake_kernels
import helpers
from helpers import batch_for, hp_for, load_golden, load_tiny_weights, rel_err
from oracle import dropout_form
from oracle.reference_form import ReferenceFormModel


class _Env:
    robot_type = None


def _ours(hp, dev):
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    reg = ModelRegistrar("/tmp", dev)
    return MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, dev, []), reg


def _zero_p(modules):
    for m in modules.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


def _batch64(mode, tiny=True, B=None):
    b = batch_for(mode, tiny, B)
    t = {k: torch.from_numpy(v) for k, v in b.items()}
    f = lambda k: t[k].double() if k in t and t[k] is not None else None
    return t, f("x"), f("x_st"), f("y"), f("y_st"), f("lanes"), f("map")


# ------------------------------------------------------------------------------------------ oracle pins (CPU)
def test_philox_known_answers():
    """Random123's kat_vectors for philox4x32_10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = fake_kernels.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want


@pytest.mark.parametrize("mode,autoregressive", [("basic", True), ("basic", False), ("map_cnn", True), ("lane_cnn", True)])
def test_mask_oracle_without_masks_is_the_reference_form(mode, autoregressive):
    """The explicit restatement must be the stock torch modules' graph: no masks == dropout probability 0."""
    z, _ = load_golden("tiny_" + mode)
    hp, B, ph = hp_for(mode, True)
    hp["autoregressive"] = autoregressive
    ref = ReferenceFormModel(hp)
    load_tiny_weights(ref.m, z)
    ref.m.double()
    ref.m.train()
    _zero_p(ref.m)
    _, x, x_st, y, y_st, lanes, maps = _batch64(mode)
    want = ref.train_loss(x, x_st, y, y_st, lanes, maps, ph, train=False)
    got = dropout_form.train_loss(ref, x, x_st, y_st, lanes, maps, ph, keep=None)
    assert abs(float(got) - float(want)) <= 1e-12 * abs(float(want))
    gw = torch.autograd.grad(want, [p for p in ref.m.parameters()], allow_unused=True)
    gg = torch.autograd.grad(got, [p for p in ref.m.parameters()], allow_unused=True)
    for a, b in zip(gw, gg):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 1e-12 * (1.0 + float(a.abs().max()))


# ------------------------------------------------------------------------------------------ placement pinned to the REFERENCE
REF_DROPOUT = [("tiny_basic_nonar_dropout", "basic", False), ("tiny_map_cnn_dropout", "map_cnn", True)]


def _ref_dropout_case(tag, mode, ar):
    """Fixture written by tests/golden/make_golden.py::run_dropout from the REAL reference's train_loss, F.dropout replaced by
    pre-drawn masks (attention-probability dropout off: inside SDPA, not injectable)."""
    z, meta = load_golden(tag)
    hp, B, ph = hp_for(mode, True)
    hp["autoregressive"] = ar
    keep = {k[len("keep::"):]: torch.from_numpy(z[k]) != 0 for k in z.files if k.startswith("keep::")}
    assert set(keep) == set(meta["sites"])
    return z, meta, hp, B, ph, keep


def _match_reference_probabilities(modules, meta, ar):
    """The state the reference was run in: attention-probability dropout off everywhere; for the autoregressive fixture the
    decoder's nn.Dropout probabilities at 0 (its passes would redraw)."""
    for key, mod in modules.items():
        for m in mod.modules():
            if isinstance(m, torch.nn.MultiheadAttention):
                m.dropout = 0.0
            if isinstance(m, torch.nn.Dropout) and ar and key.endswith("transformer_decoder"):
                m.p = 0.0


@pytest.mark.parametrize("tag,mode,ar", REF_DROPOUT)
def test_mask_oracle_places_dropout_where_the_reference_does(tag, mode, ar):
    """oracle/dropout_form.py with the reference's masks == the reference's loss and gradients: the PE,
    dropout1/2/3, feed-forward-inner and map-feature sites of the restatement are pinned to the real reference, scaling
    included.  fp64 oracle against the reference's fp32 run: 1e-5."""
    z, meta, hp, B, ph, keep = _ref_dropout_case(tag, mode, ar)
    ref = ReferenceFormModel(hp)
    load_tiny_weights(ref.m, load_golden("tiny_" + mode)[0])
    ref.m.double()
    ref.m.train()
    _match_reference_probabilities(ref.m, meta, ar)
    probs = dropout_form.site_probabilities(ref.m, hp)
    assert {k for k, p in probs.items() if p > 0} == set(keep)
    assert all(abs(probs[k] - meta["probs"][k]) < 1e-12 for k in keep)
    _, x, x_st, y, y_st, lanes, maps = _batch64(mode)
    got = dropout_form.train_loss(ref, x, x_st, y_st, lanes, maps, ph, keep=keep)
    assert abs(float(got) - float(z["train_loss"])) <= 1e-5 * abs(float(z["train_loss"]))
    got.backward()
    n = 0
    for key, mod in ref.m.items():
        for pn, p in mod.named_parameters():
            name = f"grad::{key}.{pn}"
            if name in z.files:
                assert rel_err(p.grad.numpy(), z[name]) <= 2e-5, name
                n += 1
            else:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    assert n > 40
    free = dropout_form.train_loss(ref, x, x_st, y_st, lanes, maps, ph, keep=None)
    assert abs(float(free) - float(got)) > 1e-3 * abs(float(got))          # the masks did something


def _engine_vs_reference_dropout(tag, mode, ar, dev, tol):
    from multimodal_transformer_amd import training
    z, meta, hp, B, ph, keep = _ref_dropout_case(tag, mode, ar)
    model, reg = _ours(hp, dev)
    load_tiny_weights(reg.model_dict, load_golden("tiny_" + mode)[0])
    _match_reference_probabilities(reg.model_dict, meta, ar)
    t = {k: torch.from_numpy(v) for k, v in batch_for(mode, True).items()}
    eng = training.TrainEngine(model, B, t["x_st"].shape[1], ph, seed=1)
    assert eng.drop.active and {n for n, v in eng.drop.view.items() if v is not None} == set(keep)
    eng.drop.inject(keep)
    to = lambda k: t[k].to(dev) if k in t and t[k] is not None else None
    eng.set_batch(to("x_st"), to("y_st"), to("map"), to("lanes"))
    loss = float(eng.forward())
    eng.backward()
    assert all(torch.equal(eng.drop.masks()[k].cpu().reshape(-1), keep[k].reshape(-1)) for k in keep)       # nothing redrawn
    assert abs(loss - float(z["train_loss"])) <= tol * abs(float(z["train_loss"])), (loss, float(z["train_loss"]))
    worst = 0.0
    for name, g in eng.grad_by_name.items():
        e = rel_err(g.cpu().numpy(), z["grad::" + name])
        worst = max(worst, e)
        assert e <= tol, (name, e)
    assert {"grad::" + n for n in eng.grad_by_name} == {k for k in z.files if k.startswith("grad::")}
    return worst


@pytest.mark.parametrize("tag,mode,ar", REF_DROPOUT)
def test_engine_places_dropout_where_the_reference_does_on_stand_in_kernels(monkeypatch, tag, mode, ar):
    fake_kernels.install(monkeypatch)
    print("worst gradient rel err", _engine_vs_reference_dropout(tag, mode, ar, "cpu", 1e-4))


@pytest.mark.gpu
@pytest.mark.parametrize("tag,mode,ar", REF_DROPOUT)
def test_hip_step_places_dropout_where_the_reference_does(tag, mode, ar):
    """The HIP training step fed the masks the REAL reference was run with: loss and every gradient within 1e-4 of the
    reference's.  Pins PE / dropout1-2-3 / feed-forward-inner / map-feature placement and scaling to the reference; the
    attention-probability site stays restatement-only (test_training_step_with_dropout_matches_mask_oracle)."""
    print("worst gradient rel err", _engine_vs_reference_dropout(tag, mode, ar, _gdev(), 1e-4))


# ------------------------------------------------------------------------------------------ engine vs oracle
def _engine_vs_oracle(mode, autoregressive, dev, tiny, B=None, tol=1e-4, seed=3, stash_decode=None):
    """Build the engine with the modules' own probabilities, run forward + backward, export the masks it drew, and compare
    loss and every gradient with the oracle's autograd under the same masks."""
    from multimodal_transformer_amd import training
    from multimodal_transformer_amd.weights import apply_recipe
    z = load_golden("tiny_" + mode)[0] if tiny else None
    hp, B0, ph = hp_for(mode, tiny)
    B = B or B0
    hp["autoregressive"] = autoregressive
    model, reg = _ours(hp, dev)
    ref = ReferenceFormModel(hp)
    if tiny:
        load_tiny_weights(reg.model_dict, z)
        load_tiny_weights(ref.m, z)
    else:
        apply_recipe(reg.model_dict, seed=0)
        apply_recipe(ref.m, seed=0)
    ref.m.double()
    ref.m.train()
    t, x, x_st, y, y_st, lanes, maps = _batch64(mode, tiny, B)
    eng = training.TrainEngine(model, B, t["x_st"].shape[1], ph, seed=seed)
    assert eng.drop.active
    if stash_decode is not None:
        eng.stash_decode = stash_decode
    to = lambda k: t[k].to(dev) if k in t and t[k] is not None else None
    eng.set_batch(to("x_st"), to("y_st"), to("map"), to("lanes"))
    loss = float(eng.forward())
    flips = (0, 0.0)
    if stash_decode and eng._stash_decode_ok():
        # hidden units that the decode's own stash and the all-steps pass's put on different sides of their ReLU (helpers.relu_flips):
        # none -> the element-wise tolerance applies; some (activation < 1e-5) -> the whole-tensor bound below
        other = training.TrainEngine(model, B, t["x_st"].shape[1], ph, seed=seed)
        other.stash_decode = False
        other.set_batch(to("x_st"), to("y_st"), to("map"), to("lanes"))
        other.forward()
        assert torch.equal(other.drop.pool, eng.drop.pool)
        flips = helpers.relu_flips(eng, other)
        assert flips[1] <= 1e-5, flips
        del other
    eng.backward()
    keep = {k: v.cpu() for k, v in eng.drop.masks().items()}
    probs = dropout_form.site_probabilities(ref.m, hp)
    assert set(keep) == {k for k, p in probs.items() if p > 0}
    want = dropout_form.train_loss(ref, x, x_st, y_st, lanes, maps, ph, keep=keep)
    assert abs(loss - float(want)) <= tol * abs(float(want)), (loss, float(want))
    want.backward()
    worst = 0.0
    for key, mod in ref.m.items():
        for pn, p in mod.named_parameters():
            name = f"{key}.{pn}"
            if p.grad is None:
                assert name not in eng.grad_by_name or float(eng.grad_by_name[name].abs().max()) == 0.0, name
                continue
            e = rel_err(eng.grad_by_name[name].cpu().numpy(), p.grad.numpy())
            worst = max(worst, e)
            if flips[0] == 0:
                assert e <= tol, (name, e)
            else:       # a boundary unit gated differently: one row's contribution of B * ph (whole tensor: 5e-3 at B = 8)
                w = p.grad.numpy()
                l2 = float(np.linalg.norm(eng.grad_by_name[name].cpu().numpy().astype(np.float64) - w) / (np.linalg.norm(w) + 1e-30))
                assert l2 <= 5e-3, (name, l2, flips)
    # the masks did something: the dropout-free loss differs
    free = dropout_form.train_loss(ref, x, x_st, y_st, lanes, maps, ph, keep=None)
    assert abs(float(free) - float(want)) > 1e-3 * abs(float(want))
    return worst


@pytest.mark.parametrize("mode,autoregressive", [("basic", True), ("basic", False), ("map_cnn", True), ("lane_cnn", True)])
def test_engine_with_dropout_matches_mask_oracle_on_stand_in_kernels(monkeypatch, mode, autoregressive):
    fake_kernels.install(monkeypatch)
    worst = _engine_vs_oracle(mode, autoregressive, "cpu", True, tol=2e-5)
    print("worst gradient rel err", worst)


def test_disable_dropout_gives_the_dropout_free_engine(monkeypatch):
    from multimodal_transformer_amd import training
    fake_kernels.install(monkeypatch)
    hp, B, ph = hp_for("map_cnn", True)
    model, _ = _ours(hp, "cpu")
    assert training.TrainEngine(model, B, 5, ph).drop.active
    helpers.no_dropout(model)
    eng = training.TrainEngine(model, B, 5, ph)
    assert not eng.drop.active and all(v is None for v in eng.drop.view.values())
    model2, reg2 = _ours(hp_for("basic", True)[0], "cpu")
    reg2.eval()
    assert not training.TrainEngine(model2, B, 5, ph).drop.active          # eval mode: like torch, no dropout


# ------------------------------------------------------------------------------------------ GPU
def _gdev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_mask_kernel_is_philox_bit_for_bit_and_keeps_at_the_right_rate():
    from multimodal_transformer_amd import _train_kernels as K
    dev = _gdev()
    rng = torch.tensor([0x0123456789abcdef, 5], dtype=torch.int64)
    n = 1 << 20
    for p, sid in ((0.2, 0), (0.5, 1), (0.1, 7), (0.0, 2)):
        keep = torch.empty(n, dtype=torch.uint8, device=dev)
        K.dropout_mask(keep, p, rng.to(dev), sid)
        got = keep.cpu().numpy()
        want = fake_kernels.dropout_keep_bytes(n, p, int(rng[0]), int(rng[1]), sid)
        assert np.array_equal(got, want), (p, sid)
        rate = got.mean()
        assert abs(rate - (1.0 - p)) <= 4.0 * np.sqrt(p * (1 - p) / n) + 1e-12, (p, rate)     # 4 sigma


@pytest.mark.gpu
@pytest.mark.parametrize("mode,autoregressive,tiny", [("basic", True, True), ("basic", False, True), ("map_cnn", True, True),
                                                      ("lane_cnn", True, True), ("basic", True, False), ("map_cnn", True, False),
                                                      ("lane_cnn", True, False)])
def test_training_step_with_dropout_matches_mask_oracle(mode, autoregressive, tiny):
    """Loss and every gradient of the HIP training step with dropout on (p = 0.2 Transformer / PE, 0.5 map feature) against
    the fp64 oracle run with the very masks the step drew; tolerance 1e-4 relative (north star) with the stash recomputed by the
    all-steps pass, 3e-4 with the stash the pipeline decode wrote (tag bits: test_full_config_step_matches_reference_gradient_norms;
    observed 2.0e-4 at the encoder's input_fc, the far end of the chain) unless that stash has a hidden unit on the other side of its
    ReLU (_engine_vs_oracle: then whole-tensor bounds)."""
    worst = _engine_vs_oracle(mode, autoregressive, _gdev(), tiny, B=None if tiny else 8, tol=1e-4, stash_decode=False)
    worst2 = _engine_vs_oracle(mode, autoregressive, _gdev(), tiny, B=None if tiny else 8, tol=3e-4, stash_decode=True)
    print("worst gradient rel err", worst, "with the decode's own stash", worst2)


@pytest.mark.gpu
def test_masks_are_fresh_every_step_and_reproducible_from_the_seed():
    from multimodal_transformer_amd import training
    dev = _gdev()
    hp, B, ph = hp_for("basic", True)
    hp["autoregressive"] = True
    model, reg = _ours(hp, dev)
    t = {k: torch.from_numpy(v).to(dev) for k, v in batch_for("basic", True).items()}
    tr = training.Trainer(model, B, t["x_st"].shape[1], ph, use_graph=True, seed=11)
    assert tr.engine.drop.active
    seen = []
    for _ in range(3):
        tr.step(t["x_st"], t["y_st"])
        seen.append(tr.engine.drop.pool.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])      # graph replays draw new masks
    assert int(tr.engine.drop.rng[1]) >= 3
    eng = training.TrainEngine(model, B, t["x_st"].shape[1], ph, seed=11)
    eng.seed(11, int(tr.engine.drop.rng[1]) - 1)
    eng.set_batch(t["x_st"], t["y_st"])
    eng.forward()
    assert torch.equal(eng.drop.pool, seen[2])                                         # (seed, step) determines the masks
    keep = eng.drop.masks()
    for name, m in keep.items():
        p = 0.2
        assert abs(float(m.float().mean()) - (1 - p)) < 0.08, name


@pytest.mark.gpu
def test_obtain_encoded_tensors_in_train_mode_drops_the_map_feature():
    """mgcvae.py:291-293: F.dropout(encoded_map, 0.5, training=(mode == TRAIN))."""
    from multimodal_transformer_amd.model.mgcvae import ModeKeys
    dev = _gdev()
    hp, B, ph = hp_for("map_cnn", False)
    model, reg = _ours(hp, dev)
    t = {k: torch.from_numpy(v).to(dev) for k, v in batch_for("map_cnn", False, 64).items()}
    ev = model.obtain_encoded_tensors(ModeKeys.EVAL, t["x"], t["x_st"], None, t["map"])[6]
    tr = model.obtain_encoded_tensors(ModeKeys.TRAIN, t["x"], t["x_st"], None, t["map"])[6]
    tr2 = model.obtain_encoded_tensors(ModeKeys.TRAIN, t["x"], t["x_st"], None, t["map"])[6]
    kept = tr != 0
    assert 0.4 < float(kept.float().mean()) < 0.6
    assert torch.allclose(tr[kept], 2.0 * ev[kept], rtol=1e-6, atol=0)
    assert not torch.equal(tr != 0, tr2 != 0)
    # F.dropout(..., training=(mode == TRAIN)) does not consult the module's train / eval flag: still dropped
    reg.eval()
    tr3 = model.obtain_encoded_tensors(ModeKeys.TRAIN, t["x"], t["x_st"], None, t["map"])[6]
    assert 0.4 < float((tr3 != 0).float().mean()) < 0.6
    assert torch.equal(model.obtain_encoded_tensors(ModeKeys.EVAL, t["x"], t["x_st"], None, t["map"])[6], ev)


class _FixedMask(torch.nn.Module):
    """Stands in for an nn.Dropout of the ViT restatement: multiplies by the injected keep mask / (1-p)."""

    def __init__(self, keep, p):
        super().__init__()
        self.m = keep.float() / (1.0 - p)

    def forward(self, x):
        return x * self.m.view(x.shape)


@pytest.mark.gpu
def test_vit_training_step_with_its_dropouts_matches_restatement_autograd():
    """Map_encoding_ViT with the ViT's own dropouts (0.1 after the position embedding, after to_out, inside and after the MLP)
    on and the Transformer's off.  PARITY UNPINNED (vit_pytorch is not in the reference tree): the checker is the CPU
    restatement's autograd with the nn.Dropout modules replaced by the masks the step drew."""
    import copy
    from multimodal_transformer_amd import training
    from multimodal_transformer_amd.synthetic import make_batch
    from multimodal_transformer_amd.weights import apply_recipe
    dev = _gdev()
    hp, B, ph = hp_for("map_vit", True)
    hp = copy.deepcopy(hp)
    hp["map_encoder"]["vit_param"].update(dim=32, deep=2, heads=2, mlp_dim=64)
    model, reg = _ours(hp, dev)
    apply_recipe(reg.model_dict, seed=0)
    ref = ReferenceFormModel(hp)
    apply_recipe(ref.m, seed=0)
    ref.m.train()
    for mods in (reg.model_dict, ref.m):
        for key, mod in mods.items():
            if "map_encoder" not in key:
                _zero_p(mod)
    batch = make_batch(B, hp["maximum_history_length"], ph, seed=5, with_map=True)
    c = {k: torch.from_numpy(v) for k, v in batch.items()}
    eng = training.TrainEngine(model, B, c["x_st"].shape[1], ph, seed=5)
    assert eng.drop.active and set(eng.drop.masks()) == {"vit.emb"} | {f"vit{i}.{s}" for i in range(2) for s in ("out", "ff1", "ff2")}
    eng.set_batch(c["x_st"].to(dev), c["y_st"].to(dev), maps=c["map"].to(dev))
    loss = eng.forward()
    eng.backward()
    keep = {k: v.cpu() for k, v in eng.drop.masks().items()}
    vit = ref.m["VEHICLE/map_encoder"]
    p = hp["map_encoder"]["vit_param"]["dropout"]
    vit.dropout = _FixedMask(keep["vit.emb"], hp["map_encoder"]["vit_param"]["emb_dropout"])
    for i, (attn, ff) in enumerate(vit.transformer.layers):
        attn.fn.to_out[1] = _FixedMask(keep[f"vit{i}.out"], p)
        ff.fn.net[2] = _FixedMask(keep[f"vit{i}.ff1"], p)
        ff.fn.net[4] = _FixedMask(keep[f"vit{i}.ff2"], p)
    want_loss = ref.train_loss(c["x"], c["x_st"], c["y"], c["y_st"], None, c["map"], ph)
    want_loss.backward()
    assert abs(float(loss) - float(want_loss)) <= 1e-4 * abs(float(want_loss))
    for name, prm in ref.m.named_parameters():
        if prm.grad is not None:
            assert rel_err(eng.grad_by_name[name].cpu().numpy(), prm.grad.numpy()) <= 1e-4, name


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["basic", "map_cnn", "lane_cnn", "basic-140"])
def test_masked_autoregressive_decode_equals_the_step_by_step_decode(mode):
    """With dropout the tokens fed back depend on the masks.  At the shipped geometry the inference decoder's one-launch pipeline
    and its launch-per-phase kernels read the step's keep bytes (mmt_decoder_forward_dropout); everywhere else the engine decodes
    step by step on the stash kernels.  Same seed, same masks: all three must give the same positions, loss and gradients, the
    two inference forms bit for bit."""
    from multimodal_transformer_amd import ops, training
    from multimodal_transformer_amd.weights import apply_recipe
    dev = _gdev()
    B = 24
    if mode == "basic-140":          # 129..256 agents: still 4-agent clusters (more of them than CUs / 8 in every launch)
        mode, B = "basic", 140
    hp, _, ph = hp_for(mode, False)
    hp["autoregressive"] = True
    model, reg = _ours(hp, dev)
    apply_recipe(reg.model_dict, seed=0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in batch_for(mode, False, B).items()}
    out = []
    try:
        # masked decode on the one-launch pipeline (AUTO picks it at this geometry; its DROP instantiation), on the launch-per-phase
        # kernels, and the step-by-step decode on the stash kernels
        for form in ("auto", "cluster_launches", None):
            ops.decoder_select(form or "auto")
            eng = training.TrainEngine(model, B, t["x_st"].shape[1], ph, seed=21)
            assert eng.drop.active and eng._ar_decode_takes_masks()
            eng.stash_decode = False         # all three read the all-steps pass's stash: their gradients are comparable bit for bit
            if form is None:
                eng._ar_masks_ok = False
            eng.set_batch(t["x_st"], t["y_st"], t.get("map"), t.get("lanes"))
            loss = float(eng.forward())
            if form == "auto":
                assert ops.decoder_last_path()[0] == 5      # MMT_PATH_CLUSTER_PIPELINE (every size here is within its 256 agents)
            elif form == "cluster_launches":
                assert ops.decoder_last_path()[0] == 3      # MMT_PATH_CLUSTER_LAUNCHES4
            ops.decoder_status()
            eng.backward()
            out.append((loss, eng.predictions().clone(), eng.flat_grad.clone(), eng.drop.pool.clone()))
    finally:
        ops.decoder_select("auto")
    assert torch.equal(out[0][3], out[2][3]) and torch.equal(out[1][3], out[2][3])          # the very same masks
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])   # pipeline == launches, bit for bit
    for o in out[:2]:
        assert abs(o[0] - out[2][0]) <= 1e-5 * abs(out[2][0])
        assert rel_err(o[1].cpu().numpy(), out[2][1].cpu().numpy()) <= 1e-5
        assert rel_err(o[2].cpu().numpy(), out[2][2].cpu().numpy()) <= 1e-4
