"""Deterministic by-name weight recipe (golden set 2).

Checkpoints cannot be downloaded and 32 MB of weights should not be committed,
so full-size parity fixtures are generated from a recipe that depends only on
the state_dict key, the tensor shape and a base seed.  The golden generator
applies it to the *reference* modules in the build container; tests and the
bench apply it to this package's modules.  Same key + shape + seed => same bits.
"""
from __future__ import annotations

import math
import re
import zlib

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode("utf-8")) + 1000003 * seed) % (2 ** 31 - 1))
    return g


_LN_PAT = re.compile(r"(\.norm\d*\.|\.lmlp_\d+\.1\.|\.mlp_head\.0\.)")


def _is_layernorm(key: str) -> bool:
    """norm1/2/3 of the transformer layers, Mlp's lmlp_i.1, ViT's PreNorm.norm and mlp_head.0."""
    return _LN_PAT.search(key) is not None


def recipe_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    """Value for state_dict entry ``key`` (a '/'-joined registrar key + '.' + param name)."""
    shape = tuple(shape)
    g = _gen(key, seed)
    u = torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith("pos_embedding") or key.endswith("cls_token"):
        return u * 0.5
    if _is_layernorm(key):
        # LayerNorm: weight around 1, bias around 0
        return 1.0 + 0.1 * u if leaf == "weight" else 0.1 * u
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        w = u * (1.0 / math.sqrt(fan_in))
        if "decoder/MLP" in key and "out_mlp" in key:
            w = w * 0.1      # keep per-step displacement nuScenes-sized (~0.05 standardised)
        return w
    b = 0.1 * u
    if "decoder/MLP" in key and "out_mlp" in key:
        b = b * 0.1
    return b


@torch.no_grad()
def apply_recipe(model_dict, seed: int = 0, skip=("pos_encoder.pe",)) -> None:
    """Overwrite every parameter/buffer of every module in a registrar ModuleDict."""
    for name, module in model_dict.items():
        for pname, tensor in module.state_dict().items():
            if any(pname.endswith(s) for s in skip):
                continue
            tensor.copy_(recipe_tensor(f"{name}.{pname}", tensor.shape, seed))


def checksum(model_dict) -> dict:
    """Per-module float64 sum / abs-sum: committed with the goldens to prove weight identity."""
    out = {}
    for name, module in model_dict.items():
        s = 0.0
        a = 0.0
        n = 0
        for pname, tensor in module.state_dict().items():
            if pname.endswith("pos_encoder.pe"):
                continue
            t = tensor.detach().double().cpu()
            s += float(t.sum())
            a += float(t.abs().sum())
            n += t.numel()
        out[name] = {"sum": s, "abs_sum": a, "numel": n}
    return out
