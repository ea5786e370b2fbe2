"""Generate golden vectors from the REAL reference modules (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference/Transformer/{nn_component,model} with two in-process stubs
(SURVEY.md §8c): ``vit_pytorch`` (not installed; ViT goldens are therefore NOT produced by
the reference -> ViT parity is "unpinned") and ``environment.scene_graph.DirectedEdge``
(only a name string, unused by the maths).  Nothing of the reference is copied: the
outputs written here are tensors (inputs / weights of a tiny config / outputs).

Two golden sets:
  tiny_<mode>.npz   in_dim 32, 2 heads, fdim 64, 2 layers, B=4, T=5, ph=3; full weights,
                    every module output, predict, train_loss, grads, params after 1 AdamW step.
  full_<mode>.npz   shipped sizes (d 256, 8 heads, f 512, 6+6 layers), B=128, T=9, ph=12;
                    weights from the by-name recipe (multimodal_transformer_amd/weights.py,
                    seed 0), inputs from multimodal_transformer_amd/synthetic.py (seed 1);
                    only outputs, loss, gradient norms and weight checksums are stored.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn

from multimodal_transformer_amd.config import make_hyperparams
from multimodal_transformer_amd.synthetic import make_batch
from multimodal_transformer_amd.weights import apply_recipe, checksum

REF = "/root/reference/Transformer"


def _install_stubs():
    vit = types.ModuleType("vit_pytorch")

    class ViT(nn.Module):  # placeholder; never instantiated for golden generation
        def __init__(self, **kw):
            super().__init__()
            raise RuntimeError("vit_pytorch is not available offline: ViT parity is unpinned")

    vit.ViT = ViT
    sys.modules["vit_pytorch"] = vit
    env = types.ModuleType("environment")
    env.__path__ = []
    sg = types.ModuleType("environment.scene_graph")

    class DirectedEdge:
        @staticmethod
        def get_str_from_types(a, b):
            return f"{a}->{b}"

    sg.DirectedEdge = DirectedEdge
    env.scene_graph = sg
    sys.modules["environment"] = env
    sys.modules["environment.scene_graph"] = sg
    sys.path.insert(0, REF)


_install_stubs()
from model.mgcvae import MultimodalGenerativeCVAE  # noqa: E402  (reference)
from model.model_registrar import ModelRegistrar   # noqa: E402  (reference)


class _Env:
    robot_type = None


def build_reference(hp, seed):
    torch.manual_seed(seed)
    reg = ModelRegistrar("/tmp/unused_model_dir", "cpu")
    model = MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, "cpu", [])
    return model, reg


def tensors(batch):
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    return t["x"], t["x_st"], t["y"], t["y_st"], t.get("lanes"), t.get("map")


def zero_dropout(reg, hp):
    for mod in reg.model_dict.values():
        for m in mod.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
            if isinstance(m, nn.MultiheadAttention):
                m.dropout = 0.0
    hp["map_encoder"]["cnn_param"]["dropout"] = 0.0


def run_mode(mode, tiny):
    if tiny:
        hp = make_hyperparams(mode, history=4, horizon=3,
                              transformer={"nlayers": 2, "nhead": 2, "in_dim": 32, "fdim": 64, "output_size": 32})
        B, ph = 4, 3
    else:
        hp = make_hyperparams(mode, history=8, horizon=12)
        B, ph = 128, 12
    model, reg = build_reference(hp, seed=1234)
    apply_recipe(reg.model_dict, seed=0)
    batch = make_batch(B, history=hp["maximum_history_length"], horizon=ph, seed=1,
                       with_map=(mode == "map_cnn"), with_lanes=(mode == "lane_cnn"))
    x, x_st, y, y_st, lanes, maps = tensors(batch)
    out = {}
    meta = {"mode": mode, "tiny": tiny, "B": B, "ph": ph, "torch": torch.__version__,
            "weights_checksum": checksum(reg.model_dict)}

    # ---------------- forward goldens: eval mode, no grad
    for m in reg.model_dict.values():
        m.eval()
    with torch.no_grad():
        hist, lane = model.predict(inputs=x, inputs_st=x_st, inputs_lane=lanes, lane_t_mask=None, map=maps,
                                   prediction_horizon=ph)
        (memory, smask, pmask, n0, pos, enc_lane, enc_map) = model.obtain_encoded_tensors(
            mode=None, inputs=x, inputs_st=x_st, inputs_lane=lanes, map=maps)
        eval_loss, _ = model.eval_loss("VEHICLE", inputs=x, inputs_st=x_st, inputs_lane=lanes, lane_label=None,
                                       lane_t_mask=None, labels=y, labels_st=y_st, map=maps, prediction_horizon=ph)
        model.class_input = []
    pred = lane if mode == "lane_cnn" else hist
    out["predict"] = pred.numpy()
    out["eval_loss"] = np.float32(eval_loss.item())
    out["memory"] = memory.numpy() if tiny else memory[:8].numpy()
    meta["memory_sum"] = float(memory.double().sum())
    meta["memory_abs_sum"] = float(memory.double().abs().sum())
    if enc_map is not None:
        out["enc_map"] = enc_map.numpy() if tiny else enc_map[:8].numpy()
    if enc_lane is not None:
        out["enc_lane"] = enc_lane.numpy() if tiny else enc_lane[:8].numpy()
    # ADE / FDE exactly as evaluation.py computes them per node (min over lanes for lane mode)
    pn, yn = pred.numpy(), y.numpy()
    if mode == "lane_cnn":
        err = np.linalg.norm(pn - yn[:, None], axis=-1)
        out["ade"], out["fde"] = err.mean(-1).min(-1), err[..., -1].min(-1)
    else:
        err = np.linalg.norm(pn - yn, axis=-1)
        out["ade"], out["fde"] = err.mean(-1), err[..., -1]

    if tiny:
        # teacher-forced decoder output on a fixed target (Level-1 API golden)
        from model.model_utils import generate_square_subsequent_mask
        with torch.no_grad():
            dec = reg.model_dict["VEHICLE/decoder/transformer_decoder"]
            tgt = torch.cat([torch.zeros(B, 1, 2), y_st[:, :-1]], dim=1)
            mem_for_dec = memory
            out["decoder_tf"] = dec(tgt, mem_for_dec, generate_square_subsequent_mask(ph, "cpu"), None, pmask).numpy()
            head = reg.model_dict["VEHICLE/decoder/MLP"]
            out["head_on_decoder_tf"] = head(torch.from_numpy(out["decoder_tf"])).numpy()

    # ---------------- train-step goldens: dropout forced to 0, anomaly mode off
    zero_dropout(reg, hp)
    for m in reg.model_dict.values():
        m.train()
    params = list(reg.parameters())
    opt = torch.optim.AdamW(
        [{"params": reg.get_all_but_name_match("map_encoder").parameters()},
         {"params": reg.get_name_match("map_encoder").parameters(), "lr": 0.0008}], lr=hp["learning_rate"])
    opt.zero_grad()
    loss, _ = model.train_loss(inputs=x, inputs_st=x_st, inputs_lane=lanes, lane_label=None, lane_t_mask=None,
                               labels=y, labels_st=y_st, map=maps, prediction_horizon=ph)
    loss.backward()
    out["train_loss"] = np.float32(loss.item())
    grad_norms = {}
    for name, mod in reg.model_dict.items():
        for pn_, p in mod.named_parameters():
            key = f"{name}.{pn_}"
            grad_norms[key] = None if p.grad is None else float(p.grad.double().norm())
            if tiny and p.grad is not None:
                out["grad::" + key] = p.grad.numpy().copy()
            elif p.grad is not None:     # full size: the first 256 elements of every gradient, element-wise pin
                out["gradhead::" + key] = p.grad.reshape(-1)[:256].numpy().copy()
    meta["grad_norms"] = grad_norms
    nn.utils.clip_grad_value_(reg.parameters(), hp["grad_clip"])
    opt.step()
    if tiny:
        for name, mod in reg.model_dict.items():
            for pn_, p in mod.named_parameters():
                out["post::" + f"{name}.{pn_}"] = p.detach().numpy().copy()
    else:
        meta["post_step_checksum"] = checksum(reg.model_dict)
        for name, mod in reg.model_dict.items():
            for pn_, p in mod.named_parameters():
                out["posthead::" + f"{name}.{pn_}"] = p.detach().reshape(-1)[:256].numpy().copy()

    if tiny:
        # weights BEFORE the step and inputs (regenerable, but committed so the tiny set is self-contained)
        model2, reg2 = build_reference(hp, seed=1234)
        apply_recipe(reg2.model_dict, seed=0)
        for name, mod in reg2.model_dict.items():
            for pn_, p in mod.state_dict().items():
                if pn_.endswith("pos_encoder.pe"):
                    continue
                out["w::" + f"{name}.{pn_}"] = p.numpy().copy()
        for k, v in batch.items():
            out["in::" + k] = v
    else:
        out["in_x_st_head"] = batch["x_st"][:4]

    tag = ("tiny_" if tiny else "full_") + mode
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    with open(os.path.join(HERE, tag + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(tag, "predict", out["predict"].shape, "train_loss", float(out["train_loss"]),
          "eval_loss", float(out["eval_loss"]), "ADE mean", float(out["ade"].mean()))


def run_nonar():
    """Teacher-forced training (train.py without --autoregressive -> non_at_dec, mgcvae.py:310-373), Basic, tiny config:
    hist_pred, train_loss, gradients and the parameters after one AdamW step.  Weights / inputs are those of
    tiny_basic.npz (same recipe, same batch)."""
    hp = make_hyperparams("basic", history=4, horizon=3, autoregressive=False,
                          transformer={"nlayers": 2, "nhead": 2, "in_dim": 32, "fdim": 64, "output_size": 32})
    B, ph = 4, 3
    model, reg = build_reference(hp, seed=1234)
    apply_recipe(reg.model_dict, seed=0)
    batch = make_batch(B, history=hp["maximum_history_length"], horizon=ph, seed=1)
    x, x_st, y, y_st, lanes, maps = tensors(batch)
    zero_dropout(reg, hp)
    for m in reg.model_dict.values():
        m.train()
    out, meta = {}, {"mode": "basic", "autoregressive": False, "B": B, "ph": ph, "torch": torch.__version__}
    with torch.no_grad():
        (memory, smask, pmask, n0, pos, _, _) = model.obtain_encoded_tensors(mode=None, inputs=x, inputs_st=x_st,
                                                                            inputs_lane=None, map=None)
        hist, _ = model.non_at_dec(memory, smask, pmask, y_st, n0, None, None, ph)
    out["hist_pred"] = hist.numpy()
    opt = torch.optim.AdamW(
        [{"params": reg.get_all_but_name_match("map_encoder").parameters()},
         {"params": reg.get_name_match("map_encoder").parameters(), "lr": 0.0008}], lr=hp["learning_rate"])
    opt.zero_grad()
    loss, _ = model.train_loss(inputs=x, inputs_st=x_st, inputs_lane=None, lane_label=None, lane_t_mask=None,
                               labels=y, labels_st=y_st, map=None, prediction_horizon=ph)
    loss.backward()
    out["train_loss"] = np.float32(loss.item())
    grad_norms = {}
    for name, mod in reg.model_dict.items():
        for pn_, p in mod.named_parameters():
            key = f"{name}.{pn_}"
            grad_norms[key] = None if p.grad is None else float(p.grad.double().norm())
            if p.grad is not None:
                out["grad::" + key] = p.grad.numpy().copy()
    meta["grad_norms"] = grad_norms
    nn.utils.clip_grad_value_(reg.parameters(), hp["grad_clip"])
    opt.step()
    for name, mod in reg.model_dict.items():
        for pn_, p in mod.named_parameters():
            out["post::" + f"{name}.{pn_}"] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "tiny_basic_nonar.npz"), **out)
    with open(os.path.join(HERE, "tiny_basic_nonar.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("tiny_basic_nonar train_loss", float(out["train_loss"]))


def run_dropout():
    """Dropout PLACEMENT pinned against the real reference: the reference's own ``train_loss`` in training mode
    with ``torch.nn.functional.dropout`` replaced by a function that multiplies by PRE-DRAWN keep masks (seeded generator) and
    records which module asked -- every ``nn.Dropout`` of the reference's modules calls ``F.dropout`` at run time
    (Transformer_model.py:28-30,77-79,126-129), and so does ``F.dropout(encoded_map, ...)`` at mgcvae.py:291-293.  The
    attention-probability dropout lives inside ``scaled_dot_product_attention`` where nothing can be injected: it is switched
    off (``nn.MultiheadAttention.dropout = 0``) and stays pinned to this repository's restatement only.

    Two fixtures, tiny config, weights / inputs of the tiny goldens:
      tiny_basic_nonar_dropout  teacher-forced training (ONE decoder pass: every decoder site is hit exactly once): PE (enc, dec),
                                dropout1/2 + feed-forward inner of every encoder layer, dropout1/2/3 + inner of every decoder layer
      tiny_map_cnn_dropout      autoregressive Map_encoding_CNN with the decoder's probabilities at 0 (its passes would redraw):
                                encoder sites + the map feature (p = 0.5)
    Stored: keep masks under the training engine's site names in ITS layout (encoder rows b*T+t, decoder step-major), the
    loss and every gradient."""
    import torch.nn.functional as Fn
    tiny_tr = {"nlayers": 2, "nhead": 2, "in_dim": 32, "fdim": 64, "output_size": 32}
    for tag, mode, ar in (("tiny_basic_nonar_dropout", "basic", False), ("tiny_map_cnn_dropout", "map_cnn", True)):
        hp = make_hyperparams(mode, history=4, horizon=3, autoregressive=ar, transformer=tiny_tr)
        B, ph = 4, 3
        model, reg = build_reference(hp, seed=1234)
        apply_recipe(reg.model_dict, seed=0)
        batch = make_batch(B, history=hp["maximum_history_length"], horizon=ph, seed=1, with_map=(mode == "map_cnn"))
        x, x_st, y, y_st, lanes, maps = tensors(batch)
        site_of = {}
        for key, mod in reg.model_dict.items():
            side = "enc" if key.endswith("node_history_encoder") else ("dec" if key.endswith("transformer_decoder") else None)
            for name, m in mod.named_modules():
                if isinstance(m, nn.MultiheadAttention):
                    m.dropout = 0.0                    # (inside SDPA: cannot be injected)
                if not isinstance(m, nn.Dropout):
                    continue
                if side is None:
                    raise RuntimeError(f"unexpected nn.Dropout {key}.{name}")
                if side == "dec" and ar:
                    m.p = 0.0                          # the AR loop re-runs the decoder: fresh masks per pass, not injectable once
                parts = name.split(".")
                if name == "pos_encoder.dropout":
                    site = f"{side}.pe"
                else:
                    l, leaf = int(parts[2]), parts[3]
                    site = f"{side}{l}." + {"dropout": "ff", "dropout1": "drop1", "dropout2": "drop2", "dropout3": "drop3"}[leaf]
                site_of[id(m)] = site
        current = [None]
        for key, mod in reg.model_dict.items():
            for m in mod.modules():
                if isinstance(m, nn.Dropout):
                    m.register_forward_pre_hook(lambda mm, inp: current.__setitem__(0, site_of[id(mm)]))
        gen = torch.Generator().manual_seed(4321)
        drawn, probs = {}, {}

        def fake_dropout(input, p=0.5, training=True, inplace=False):
            site, current[0] = current[0], None
            if site is None:
                site = "map.feature"                   # the only bare F.dropout call on the path (mgcvae.py:291-293)
            if not training or p == 0.0:
                return input
            if site in drawn:
                raise RuntimeError(f"dropout site {site} hit twice")
            keep = torch.rand(input.shape, generator=gen) >= p
            drawn[site], probs[site] = keep, float(p)
            return input * keep.to(input.dtype) / (1.0 - p)

        for m in reg.model_dict.values():
            m.train()
        real = Fn.dropout
        Fn.dropout = fake_dropout
        try:
            loss, _ = model.train_loss(inputs=x, inputs_st=x_st, inputs_lane=lanes, lane_label=None, lane_t_mask=None,
                                       labels=y, labels_st=y_st, map=maps, prediction_horizon=ph)
            loss.backward()
        finally:
            Fn.dropout = real
        out = {"train_loss": np.float32(loss.item())}
        for site, keep in drawn.items():
            if site.startswith("dec"):                 # (B, ph, .) -> the engine's step-major (ph, B, .)
                keep = keep.permute(1, 0, 2)
            elif site.startswith("enc"):               # (B, T, .) -> rows b*T + t
                keep = keep.reshape(-1, keep.shape[-1])
            out["keep::" + site] = keep.contiguous().numpy().astype(np.uint8)
        for name, mod in reg.model_dict.items():
            for pn_, p in mod.named_parameters():
                if p.grad is not None:
                    out["grad::" + f"{name}.{pn_}"] = p.grad.numpy().copy()
        meta = {"mode": mode, "autoregressive": ar, "B": B, "ph": ph, "torch": torch.__version__, "probs": probs,
                "sites": sorted(drawn), "attention_probability_dropout": "off in the reference run (inside SDPA, not injectable)"}
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
        with open(os.path.join(HERE, tag + ".json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(tag, "train_loss", float(out["train_loss"]), "sites", sorted(drawn))


def run_mlp():
    """nn_component.MLP.Mlp stand-alone: regression and classification (nn.Softmax() with dim=None: dim 0 for the 3-D
    input the reference gives it, dim 1 for a 2-D one)."""
    from nn_component.MLP import Mlp          # reference
    import warnings
    out = {}
    for tag, mode, nin, nout, layers in (("cls", "classification", 96, 3, 2), ("reg", "regression", 32, 2, 3)):
        torch.manual_seed(7)
        m = Mlp(nin, nout, layers, mode)
        g = torch.Generator().manual_seed(11)
        x3, x2 = torch.randn(4, 3, nin, generator=g), torch.randn(5, nin, generator=g)
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[f"{tag}::y3"], out[f"{tag}::y2"] = m(x3).numpy(), m(x2).numpy()
        out[f"{tag}::x3"], out[f"{tag}::x2"] = x3.numpy(), x2.numpy()
        for k, v in m.state_dict().items():
            out[f"{tag}::w::{k}"] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "mlp_standalone.npz"), **out)
    print("mlp_standalone", {k: v.shape for k, v in out.items() if "::y" in k})


def run_gmm2d():
    """Transformer/model/components/gmm2d.py (dead code in the live path, named by the north star): attributes,
    log_prob, reg_los, covariance, from_log_pis_mus_cov_mats, mode (0.01 grid search) and rsample with the normal
    noise / component draw pinned (torch.randn and Categorical.sample are replaced by stored tensors)."""
    from model.components.gmm2d import GMM2D      # reference
    import torch.distributions as td
    g = torch.Generator().manual_seed(5)
    S, Bn, Tn, N = 1, 3, 2, 4
    log_pis = torch.randn(S, Bn, Tn, N, generator=g)
    log_pis[0, 0, 0, 1] = -2e5                                     # exercises the clamp at -1e5
    mus = torch.randn(S, Bn, Tn, N * 2, generator=g) * 0.15        # small spread: mode() grids of <= ~60 x 60 points
    log_sigmas = torch.randn(S, Bn, Tn, N * 2, generator=g) * 0.3 - 1.5
    corrs = torch.tanh(torch.randn(S, Bn, Tn, N, generator=g))
    corrs[0, 1, 1, 2] = 0.9999999                                  # exercises the clamp of 1 - rho^2
    value = torch.randn(S, Bn, Tn, 2, generator=g) * 0.2
    gm = GMM2D(log_pis, mus, log_sigmas, corrs)
    out = {"log_pis": log_pis.numpy(), "mus": mus.numpy(), "log_sigmas": log_sigmas.numpy(), "corrs": corrs.numpy(),
           "value": value.numpy(), "norm_log_pis": gm.log_pis.numpy(), "sigmas": gm.sigmas.numpy(),
           "one_minus_rho2": gm.one_minus_rho2.numpy(), "L": gm.L.numpy(), "log_prob": gm.log_prob(value).numpy(),
           "reg_los": gm.reg_los(value).numpy(), "cov": gm.get_covariance_matrix().numpy(), "mode": gm.mode().numpy()}
    gm2 = GMM2D.from_log_pis_mus_cov_mats(log_pis, mus, gm.get_covariance_matrix())
    out["from_cov_log_sigmas"], out["from_cov_corrs"] = gm2.log_sigmas.numpy(), gm2.corrs.numpy()
    # rsample with pinned randomness, sample_shape (5,)
    z = torch.randn(5, S, Bn, Tn, N, 2, generator=g)
    comp = torch.randint(0, N, (5, S, Bn, Tn), generator=g)
    real_randn, real_sample = torch.randn, td.Categorical.sample
    try:
        torch.randn = lambda size, device=None: z
        td.Categorical.sample = lambda self, sample_shape=torch.Size(): comp
        out["rsample"] = gm.rsample(torch.Size([5])).numpy()
    finally:
        torch.randn, td.Categorical.sample = real_randn, real_sample
    out["rsample_z"], out["rsample_comp"] = z.numpy(), comp.numpy()
    # single component: mode() returns the mean
    g1 = GMM2D(log_pis[..., :1], mus[..., :2], log_sigmas[..., :2], corrs[..., :1])
    out["mode_single"] = g1.mode().numpy()
    np.savez_compressed(os.path.join(HERE, "gmm2d.npz"), **out)
    print("gmm2d", {k: v.shape for k, v in out.items()})


def run_mapcrop():
    """environment/map.py GeometricMap.get_cropped_maps_from_scene_map_batch (pad, crop, rotate with bilinear
    grid_sample, slice) on two synthetic uint8 rasters: agents in the interior and at the borders (zero padding), with and without rotation, two patch sizes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_env_map", os.path.join(REF, "environment", "map.py"))
    refmap = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(refmap)                   # reference
    rng = np.random.default_rng(3)

    def raster(X, Y):
        base = rng.random((3, X // 8 + 2, Y // 8 + 2)) > 0.55            # blobs of ~8 px (a drivable-area-like mask)
        up = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)[:, :X, :Y]
        img = up.astype(np.uint8) * 255
        img[1] = (rng.random((X, Y)) * 255).astype(np.uint8)             # one layer of per-pixel noise: hardest for bilinear parity
        return img

    maps = [refmap.GeometricMap(raster(420, 600), np.array([[3., 0., 12.], [0., 3., 7.], [0., 0., 1.]])),
            refmap.GeometricMap(raster(333, 257), np.array([[2.5, 0., -20.], [0., 2.5, 31.], [0., 0., 1.]]))]
    out = {"raster0": maps[0].data, "raster1": maps[1].data, "homography0": maps[0].homography, "homography1": maps[1].homography}
    pts = np.array([[60.3, 90.7], [1.2, 2.9], [135.0, 195.5], [70.0, 10.0], [20.4, 40.6], [100.9, 60.1], [120.0, 80.0],
                    [135.9, 195.9]])     # (the reference fails on agents outside their map: its crop slices come out short)
    which = np.array([0, 0, 0, 0, 1, 1, 1, 0])
    rot = torch.tensor([0., 33.3, -120.0, 90.0, 179.0, -15.5, 45.0, 271.25])
    out["scene_pts"], out["which"], out["rotation"] = pts, which, rot.numpy()
    for tag, ps in (("ref", [50, 10, 50, 90]), ("small", [8, 4, 6, 12])):
        ml = [maps[w] for w in which]
        out[f"{tag}::patch_size"] = np.array(ps)
        out[f"{tag}::rotated"] = refmap.GeometricMap.get_cropped_maps_from_scene_map_batch(ml, torch.tensor(pts), ps, rotation=rot).numpy()
        out[f"{tag}::unrotated"] = refmap.GeometricMap.get_cropped_maps_from_scene_map_batch(ml, torch.tensor(pts), ps, rotation=None).numpy()
    np.savez_compressed(os.path.join(HERE, "mapcrop.npz"), **out)
    print("mapcrop", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    torch.set_num_threads(8)
    what = sys.argv[1:] or ["modes", "nonar", "dropout", "mlp", "gmm2d", "mapcrop"]
    if "modes" in what:
        for tiny in (True, False):
            for mode in ("basic", "map_cnn", "lane_cnn"):
                run_mode(mode, tiny)
    if "nonar" in what:
        run_nonar()
    if "dropout" in what:
        run_dropout()
    if "mlp" in what:
        run_mlp()
    if "gmm2d" in what:
        run_gmm2d()
    if "mapcrop" in what:
        run_mapcrop()
