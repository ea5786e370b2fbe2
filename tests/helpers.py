"""Shared helpers for the parity tests (oracle side lives under oracle/, test-only)."""
import json
import os

import numpy as np
import torch

from multimodal_transformer_amd.config import make_hyperparams
from multimodal_transformer_amd.synthetic import make_batch
from multimodal_transformer_amd.weights import apply_recipe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY_TR = {"nlayers": 2, "nhead": 2, "in_dim": 32, "fdim": 64, "output_size": 32}
MODES = ("basic", "map_cnn", "lane_cnn")


def no_dropout(model):
    """The goldens were generated with every dropout probability forced to 0: same here."""
    from multimodal_transformer_amd.training import disable_dropout
    return disable_dropout(model)


def load_golden(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    with open(os.path.join(GOLDEN, tag + ".json")) as f:
        meta = json.load(f)
    return z, meta


def hp_for(mode, tiny):
    if tiny:
        return make_hyperparams(mode, history=4, horizon=3, transformer=TINY_TR), 4, 3
    return make_hyperparams(mode, history=8, horizon=12), 128, 12


def batch_for(mode, tiny, B=None):
    hp, b, ph = hp_for(mode, tiny)
    b = B or b
    return make_batch(b, history=hp["maximum_history_length"], horizon=ph, seed=1,
                      with_map=(mode in ("map_cnn", "map_vit")), with_lanes=(mode == "lane_cnn"))


def load_tiny_weights(model_dict, z):
    """Copy the committed tiny-config weights ('w::<registrar key>.<param>') into a ModuleDict."""
    with torch.no_grad():
        for name, mod in model_dict.items():
            for pn, t in mod.state_dict().items():
                if pn.endswith("pos_encoder.pe"):
                    continue
                t.copy_(torch.from_numpy(z["w::" + f"{name}.{pn}"]))


def set_weights(model_dict, tiny, z):
    if tiny:
        load_tiny_weights(model_dict, z)
    else:
        apply_recipe(model_dict, seed=0)


STRICT_LOG = []          # (worst strict per-output relative error, where) of every rel_err call: printed at the end of the session


def rel_err(a, b, floor=None):
    """max |a-b| / max(|b|, floor) with floor defaulting to the RMS of b (per-output relative error).  The strict figure -- the
    same quotient with the floor at 1e-3 of the RMS, i.e. small-magnitude outputs judged against themselves -- is recorded in
    STRICT_LOG and summarised by conftest's terminal report ('the strict figure should be printed too')."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    rms = float(np.sqrt(np.mean(b * b))) + 1e-30
    if floor is None:
        floor = rms
    if a.size:
        import inspect
        strict = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * rms)))
        fr = inspect.stack()[1]
        STRICT_LOG.append((strict, f"{fr.filename.rsplit('/', 1)[-1]}:{fr.lineno}"))
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def check_post_step(p_new, post_ref, grad_ref, lr, tol=1e-4):
    """Post-AdamW parameters against the reference's.  The first AdamW step moves every element by
    lr * g / (|g| + 1e-8): where the reference gradient is numerically zero (|g| < 1e-7, e.g. the key bias of an
    attention layer, whose true gradient is 0 by softmax shift invariance) the update is the sign of rounding
    noise and may legitimately differ by up to 2*lr; everywhere else the parameters must agree to ``tol``."""
    p_new, post_ref = np.asarray(p_new, np.float64), np.asarray(post_ref, np.float64)
    if grad_ref is None:
        return float(np.max(np.abs(p_new - post_ref))) == 0.0, 0.0
    solid = np.abs(np.asarray(grad_ref)) >= 1e-7
    floor = float(np.sqrt(np.mean(post_ref * post_ref))) + 1e-30
    err = np.abs(p_new - post_ref)
    ok_solid = (not solid.any()) or float(np.max(err[solid] / np.maximum(np.abs(post_ref[solid]), floor))) <= tol
    ok_noise = solid.all() or float(np.max(err[~solid])) <= 2.1 * lr
    return ok_solid and ok_noise, float(err.max())


def relu_flips(eng_a, eng_b):
    """Hidden units (decoder feed-forward, decoder/MLP hidden layer) that are active in one engine's forward stash and inactive in
    the other's: (count, largest activation among them).  Two fp32 forwards that differ by rounding put a unit whose pre-activation
    is within rounding of 0 on different sides of the ReLU; the backward then gates one (agent, step) contribution of one weight
    row differently - a discontinuity of the function, not an error of either evaluation.  The training tests use this to tell
    that case from a wrong gradient: with no flipped unit the strict element-wise tolerance applies."""
    import torch
    n, mx = 0, 0.0
    pairs = [(a, b) for a, b in zip(eng_a.s.f1, eng_b.s.f1)] + [(a, b) for a, b in zip(eng_a.head_tape.hu, eng_b.head_tape.hu)]
    for a, b in pairs:
        f = (a > 0) != (b > 0)
        k = int(f.sum())
        if k:
            n += k
            mx = max(mx, float(torch.maximum(a[f].abs().max(), b[f].abs().max())))
    return n, mx
