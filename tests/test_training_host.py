"""CPU tier: the host orchestration of the training step (multimodal_transformer_amd/training.py) against the
reference's autograd goldens.  The HIP kernels are replaced by tests/fake_kernels.py (torch stand-ins with the
same argument conventions), so what is pinned here is the *graph*: forward-with-stash, the reverse-time sweep
through the KV-cached decoder, the deferred weight-gradient GEMMs, clip + AdamW + LR schedule.
The kernels themselves are pinned on the GPU by tests/test_gpu_training.py."""
import numpy as np
import pytest
import torch

import fake_kernels
import helpers
from helpers import batch_for, check_post_step, hp_for, load_golden, rel_err, set_weights

GRAD_TOL = 1e-4


class _Env:
    robot_type = None


def _model(hp):
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    reg = ModelRegistrar("/tmp", "cpu")
    return helpers.no_dropout(MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, "cpu", [])), reg


def _tiny_engine(monkeypatch):
    from multimodal_transformer_amd import training
    fake_kernels.install(monkeypatch)
    z, meta = load_golden("tiny_basic")
    hp, B, ph = hp_for("basic", True)
    hp["autoregressive"] = True
    model, reg = _model(hp)
    set_weights(reg.model_dict, True, z)
    batch = batch_for("basic", True)
    x_st, y_st = torch.from_numpy(batch["x_st"]), torch.from_numpy(batch["y_st"])
    return training, model, reg, z, meta, hp, x_st, y_st, ph


def test_forward_backward_match_reference_autograd(monkeypatch):
    training, model, reg, z, meta, hp, x_st, y_st, ph = _tiny_engine(monkeypatch)
    eng = training.TrainEngine(model, x_st.shape[0], x_st.shape[1], ph)
    eng.set_batch(x_st, y_st)
    loss = eng.forward()
    assert abs(float(loss) - float(z["train_loss"])) <= 1e-5 * abs(float(z["train_loss"]))
    eng.backward()
    # exactly the parameters the reference gives a gradient to
    want = {k for k, v in meta["grad_norms"].items() if v is not None}
    assert set(eng.grad_by_name) == want
    worst = 0.0
    for name, g in eng.grad_by_name.items():
        e = rel_err(g.numpy(), z["grad::" + name])
        worst = max(worst, e)
        assert e <= GRAD_TOL, (name, e)
    print("worst gradient rel err", worst)


def test_map_cnn_forward_backward_and_step_match_reference(monkeypatch):
    """Map_encoding_CNN (BASELINE config 3 as a training configuration): the CNN map encoder (convs as im2col + GEMM) and
    fusion/hist_map receive gradients too, the map encoder steps with its own learning rate (train.py:218-219)."""
    from multimodal_transformer_amd import training
    fake_kernels.install(monkeypatch)
    z, meta = load_golden("tiny_map_cnn")
    hp, B, ph = hp_for("map_cnn", True)
    model, reg = _model(hp)
    set_weights(reg.model_dict, True, z)
    batch = batch_for("map_cnn", True)
    x_st, y_st, maps = (torch.from_numpy(batch[k]) for k in ("x_st", "y_st", "map"))
    tr = training.Trainer(model, x_st.shape[0], x_st.shape[1], ph, use_graph=False)
    eng = tr.engine
    eng.set_batch(x_st, y_st, maps)
    loss = eng.forward()
    assert abs(float(loss) - float(z["train_loss"])) <= 1e-5 * abs(float(z["train_loss"]))
    eng.backward()
    want = {k for k, v in meta["grad_norms"].items() if v is not None}
    assert set(eng.grad_by_name) == want and any("map_encoder" in k for k in want) and any("fusion/hist_map" in k for k in want)
    for name, g in eng.grad_by_name.items():
        e = rel_err(g.numpy(), z["grad::" + name])
        assert e <= GRAD_TOL, (name, e)
    tr.step(x_st, y_st, maps)
    for name, mod in reg.model_dict.items():
        for pn, p in mod.named_parameters():
            key = f"{name}.{pn}"
            ok, e = check_post_step(p.detach().numpy(), z["post::" + key], z["grad::" + key] if "grad::" + key in z.files
                                    else None, 0.0008 if "map_encoder" in key else hp["learning_rate"])
            assert ok, (key, e)


def test_lane_cnn_forward_backward_and_step_match_reference(monkeypatch):
    """Lane_encoding_CNN as a training configuration: lane encoder (Conv1d + fc) and fusion/lane_hist receive gradients;
    Lane/MLP_Softmax does not (its output is dead in the reference, mgcvae.py:552-557)."""
    from multimodal_transformer_amd import training
    fake_kernels.install(monkeypatch)
    z, meta = load_golden("tiny_lane_cnn")
    hp, B, ph = hp_for("lane_cnn", True)
    model, reg = _model(hp)
    set_weights(reg.model_dict, True, z)
    batch = batch_for("lane_cnn", True)
    x_st, y_st, lanes = (torch.from_numpy(batch[k]) for k in ("x_st", "y_st", "lanes"))
    tr = training.Trainer(model, x_st.shape[0], x_st.shape[1], ph, use_graph=False)
    eng = tr.engine
    eng.set_batch(x_st, y_st, lanes=lanes)
    loss = eng.forward()
    assert abs(float(loss) - float(z["train_loss"])) <= 1e-5 * abs(float(z["train_loss"]))
    eng.backward()
    want = {k for k, v in meta["grad_norms"].items() if v is not None}
    assert set(eng.grad_by_name) == want and any("lane_encoder" in k for k in want)
    for name, g in eng.grad_by_name.items():
        e = rel_err(g.numpy(), z["grad::" + name])
        assert e <= GRAD_TOL, (name, e)
    tr.step(x_st, y_st, lanes=lanes)
    for name, mod in reg.model_dict.items():
        for pn, p in mod.named_parameters():
            key = f"{name}.{pn}"
            ok, e = check_post_step(p.detach().numpy(), z["post::" + key], z["grad::" + key] if "grad::" + key in z.files
                                    else None, hp["learning_rate"])
            assert ok, (key, e)


def test_fused_step_matches_reference_adamw(monkeypatch):
    training, model, reg, z, meta, hp, x_st, y_st, ph = _tiny_engine(monkeypatch)
    tr = training.Trainer(model, x_st.shape[0], x_st.shape[1], ph, use_graph=False)
    loss = tr.step(x_st, y_st)
    assert abs(float(loss) - float(z["train_loss"])) <= 1e-5 * abs(float(z["train_loss"]))
    for name, mod in reg.model_dict.items():
        for pn, p in mod.named_parameters():
            key = f"{name}.{pn}"
            ok, e = check_post_step(p.detach().numpy(), z["post::" + key], z["grad::" + key] if "grad::" + key in z.files
                                    else None, hp["learning_rate"])
            assert ok, (key, e)
    assert float(tr.state[0]) == 2.0
    assert abs(float(tr.state[1]) - hp["learning_rate"] * hp["learning_decay_rate"]) < 1e-10


def test_autograd_entry_point_drives_reference_style_loop(monkeypatch):
    """loss.backward(); clip_grad_value_; torch AdamW.step() -- the reference's own loop (train.py:244-253)."""
    training, model, reg, z, meta, hp, x_st, y_st, ph = _tiny_engine(monkeypatch)
    opt = torch.optim.AdamW(
        [{"params": reg.get_all_but_name_match("map_encoder").parameters()},
         {"params": reg.get_name_match("map_encoder").parameters(), "lr": 0.0008}], lr=hp["learning_rate"])
    opt.zero_grad()
    batch = batch_for("basic", True)
    x, y = torch.from_numpy(batch["x"]), torch.from_numpy(batch["y"])
    loss, reg_loss = model.train_loss(x, x_st, None, None, None, y, y_st, None, ph)
    assert loss.requires_grad
    loss.backward()
    for name, mod in reg.model_dict.items():
        for pn, p in mod.named_parameters():
            key = f"{name}.{pn}"
            if meta["grad_norms"][key] is None:
                assert p.grad is None, key
            else:
                assert rel_err(p.grad.numpy(), z["grad::" + key]) <= GRAD_TOL, key
    torch.nn.utils.clip_grad_value_(reg.parameters(), hp["grad_clip"])
    opt.step()
    for name, mod in reg.model_dict.items():
        for pn, p in mod.named_parameters():
            key = f"{name}.{pn}"
            ok, e = check_post_step(p.detach().numpy(), z["post::" + key], z["grad::" + key] if "grad::" + key in z.files
                                    else None, hp["learning_rate"])
            assert ok, (key, e)


def test_ragged_shape_gradients_match_torch_autograd_of_the_oracle(monkeypatch):
    """Shapes the goldens do not cover (B=3, T=7, ph=5): the engine's gradients against torch autograd run on the
    explicit-arithmetic fp64 oracle."""
    from oracle import restatement as rs
    from multimodal_transformer_amd import training
    from multimodal_transformer_amd.config import make_hyperparams
    from multimodal_transformer_amd.synthetic import make_batch
    from multimodal_transformer_amd.weights import apply_recipe
    from helpers import TINY_TR
    fake_kernels.install(monkeypatch)
    hp = make_hyperparams("basic", history=6, horizon=5, transformer=TINY_TR)
    hp["autoregressive"] = True
    model, reg = _model(hp)
    apply_recipe(reg.model_dict, seed=7)
    batch = make_batch(3, 6, 5, seed=4)
    x_st, y_st = torch.from_numpy(batch["x_st"]), torch.from_numpy(batch["y_st"])
    eng = training.TrainEngine(model, 3, 7, 5)
    eng.set_batch(x_st, y_st)
    loss = eng.forward()
    eng.backward()
    w = rs.flatten_modules(reg.model_dict)
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items() if v.is_floating_point() and not k.endswith(".pe")}
    w2 = dict(w)
    w2.update(leaves)
    memory, pad = rs.encoder_forward(w2, x_st.double(), hp["transformer"]["nhead"])
    pos, _ = rs.decoder_ar(w2, memory, pad, x_st[:, -1, 0:2].double(), hp["transformer"]["nhead"], 5)
    ref = (pos - y_st.double()).norm(dim=-1).mean(-1).mean()
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    alias = {"VEHICLE/node_history_encoder": "enc", "VEHICLE/decoder/transformer_decoder": "dec",
             "VEHICLE/decoder/MLP": "head"}
    for name, g in eng.grad_by_name.items():
        key, pn = name.rsplit("/", 1)[0] + "/" + name.rsplit("/", 1)[1].split(".", 1)[0], name.rsplit("/", 1)[1].split(".", 1)[1]
        want = leaves[f"{alias[key]}.{pn}"].grad
        assert rel_err(g.numpy(), want.numpy()) <= GRAD_TOL, name


def test_teacher_forced_step_matches_reference_non_at_dec(monkeypatch):
    """train.py without --autoregressive: non_at_dec (mgcvae.py:310-373).  Loss, gradients and the parameters after
    one AdamW step against the reference's autograd (tests/golden/tiny_basic_nonar.npz)."""
    from multimodal_transformer_amd import training
    fake_kernels.install(monkeypatch)
    z0, _ = load_golden("tiny_basic")
    z, meta = load_golden("tiny_basic_nonar")
    hp, B, ph = hp_for("basic", True)
    hp["autoregressive"] = False
    model, reg = _model(hp)
    set_weights(reg.model_dict, True, z0)
    batch = batch_for("basic", True)
    x_st, y_st = torch.from_numpy(batch["x_st"]), torch.from_numpy(batch["y_st"])
    tr = training.Trainer(model, B, x_st.shape[1], ph, use_graph=False)
    assert tr.engine.teacher_forced
    loss = tr.step(x_st, y_st)
    assert abs(float(loss) - float(z["train_loss"])) <= 1e-5 * abs(float(z["train_loss"]))
    assert rel_err(tr.engine.predictions().numpy(), z["hist_pred"]) <= 1e-5
    want = {k for k, v in meta["grad_norms"].items() if v is not None}
    assert set(tr.engine.grad_by_name) == want
    for name, g in tr.engine.grad_by_name.items():
        assert rel_err(g.numpy(), z["grad::" + name]) <= GRAD_TOL, name
    for name, mod in reg.model_dict.items():
        for pn, p in mod.named_parameters():
            key = f"{name}.{pn}"
            ok, e = check_post_step(p.detach().numpy(), z["post::" + key], z["grad::" + key] if "grad::" + key in z.files
                                    else None, hp["learning_rate"])
            assert ok, (key, e)


@pytest.mark.parametrize("mode", ["basic", "map_cnn", "lane_cnn"])
def test_gradient_buckets_partition_the_flat_buffer_by_completion_time(monkeypatch, mode):
    """The overlapped all-reduce (Trainer, world > 1) sends ``early_range`` while ``backward_late`` still runs: the two sets of
    ranges must tile the flat buffer, every history-encoder / lane-branch gradient must lie in a late range, and
    backward_early() + backward_late() must leave exactly what backward() leaves."""
    from multimodal_transformer_amd import training
    fake_kernels.install(monkeypatch)
    hp, B, ph = hp_for(mode, True)
    hp["autoregressive"] = True
    model, reg = _model(hp)
    z, _ = load_golden("tiny_" + mode)
    set_weights(reg.model_dict, True, z)
    batch = batch_for(mode, True)
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    eng = training.TrainEngine(model, t["x_st"].shape[0], t["x_st"].shape[1], ph)
    lo, hi = eng.early_range
    spans = sorted([(lo, hi)] + list(eng.late_ranges))
    assert spans[0][0] == 0 and spans[-1][1] == eng.grad_total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    base = eng.flat_grad.data_ptr()
    for name, g in eng.grad_by_name.items():
        off = (g.data_ptr() - base) // 4
        late = any(a <= off < b for a, b in eng.late_ranges)
        assert late == (("node_history_encoder" in name) or ("lane" in name)), name
    eng.set_batch(t["x_st"], t["y_st"], t.get("map"), t.get("lanes"))
    eng.forward()
    eng.backward()
    whole = eng.flat_grad.clone()
    eng.forward()
    eng.backward_early()
    early = eng.flat_grad[lo:hi].clone()
    eng.backward_late()
    assert torch.equal(early, eng.flat_grad[lo:hi])          # the late stretch does not touch the early bucket
    assert torch.equal(whole, eng.flat_grad)


def test_group_launch_tables_live_as_long_as_their_owner(monkeypatch):
    """a captured training graph holds the ADDRESS of the device launch table of every group launch.  An
    owner's cache (TrainEngine._group_cache) therefore never evicts: after > 64 other job lists have gone through the shared,
    bounded cache and through other owners, the first owner still finds its original table object.  Host-only: the plan / run
    entry points are stand-ins, the table stays on the CPU."""
    import ctypes
    from multimodal_transformer_amd import _lib
    from multimodal_transformer_amd import _train_kernels as K

    launched = []

    def plan(jobs, n, host, cap, nbytes, nblocks):
        nbytes._obj.value, nblocks._obj.value = 16 * n, n
        return 0

    def run(table, njobs, nblocks, stream):
        launched.append(table.value)
        return 0

    class _FakeLib:
        pass
    monkeypatch.setattr(_lib, "load", lambda: _FakeLib())
    monkeypatch.setattr(K, "_group_device", lambda: "cpu")
    monkeypatch.setattr(K, "_stream", lambda: None)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)

    def joblist(tag, n=3):
        jobs = (_lib.LnGradJob * n)()
        for i, J in enumerate(jobs):
            J.dy, J.M, J.N = 4096 * tag + i, 512, 256
        return jobs

    owner_a, owner_b = {}, {}
    first = K._group_launch("ln", joblist(1), plan, run, "fake", cache=owner_a)
    addr = first.data_ptr()
    for tag in range(2, 2 + 3 * K._GROUP_TABLES_MAX):          # far more job lists than the shared cache holds
        K._group_launch("ln", joblist(tag), plan, run, "fake")                     # ad-hoc, shared + bounded
        K._group_launch("ln", joblist(tag), plan, run, "fake", cache=owner_b)      # another engine
        K._group_launch("ln", joblist(tag), plan, run, "fake", cache=owner_a)      # the same engine, other shapes
    assert len(K._GROUP_TABLES) <= K._GROUP_TABLES_MAX
    assert len(owner_a) == 1 + 3 * K._GROUP_TABLES_MAX           # nothing of an owner's is ever evicted
    again = K._group_launch("ln", joblist(1), plan, run, "fake", cache=owner_a)
    assert again is first and again.data_ptr() == addr and launched[-1] == addr
    # a capture may not lean on the shared cache: its entries can be evicted under a live graph
    K._group_launch("ln", joblist(7), plan, run, "fake")
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    with pytest.raises(_lib.MmtError):
        K._group_launch("ln", joblist(7), plan, run, "fake")
    K._group_launch("ln", joblist(1), plan, run, "fake", cache=owner_a)            # an owner's table replays under capture


def test_train_engine_owns_its_group_tables(monkeypatch):
    from multimodal_transformer_amd import _train_kernels as K
    training, model, reg, z, meta, hp, x_st, y_st, ph = _tiny_engine(monkeypatch)
    seen = []
    monkeypatch.setattr(K, "gemm_group", lambda calls, cache=None: seen.append(cache))
    monkeypatch.setattr(K, "ln_param_grads_group", lambda calls, cache=None: seen.append(cache))
    eng = training.TrainEngine(model, x_st.shape[0], x_st.shape[1], ph)
    eng._deferred = ([], [])
    eng._flush_wgrads()
    assert seen and all(c is eng._group_cache for c in seen)
