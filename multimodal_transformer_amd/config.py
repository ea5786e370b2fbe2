"""Hyper-parameter dictionaries for the hot path.

The reference merges ``experiments/nuScene/config/config.json`` with a handful of
argparse booleans into one ``hyperparams`` dict (reference ``Transformer/train.py:29-45``)
and hands that dict to ``MultimodalGenerativeCVAE`` (``Transformer/model/mgcvae.py:22-62``).
This module builds the same dict from code so that tests, the bench and the
golden generator agree on one source of truth.  Only keys read by the hot path
are present.
"""
from __future__ import annotations

import copy

MODES = ("basic", "map_cnn", "map_vit", "lane_cnn")

# values of experiments/nuScene/config/config.json (reference), restated as data
_BASE = {
    "batch_size": 256,
    "grad_clip": 6.0,
    "learning_rate_style": "exp",
    "learning_rate": 0.0004,
    "min_learning_rate": 1e-05,
    "learning_decay_rate": 0.9999,
    "prediction_horizon": 12,
    "minimum_history_length": 8,
    "maximum_history_length": 8,
    "map_encoder": {
        "transformer": True,
        "vit_param": {"image_size": 100, "patch_size": 25, "output_size": 32, "dim": 256,
                      "deep": 6, "heads": 8, "mlp_dim": 512, "dropout": 0.1, "emb_dropout": 0.1},
        "cnn_param": {"heading_state_index": 6, "patch_size": [50, 10, 50, 90], "map_channels": 3,
                      "hidden_channels": [10, 20, 10, 1], "output_size": 32, "masks": [5, 5, 5, 3],
                      "strides": [2, 2, 1, 1], "dropout": 0.5},
    },
    "state": {
        "PEDESTRIAN": {"position": ["x", "y"], "velocity": ["x", "y"], "acceleration": ["x", "y"]},
        "VEHICLE": {"position": ["x", "y"], "velocity": ["x", "y"], "acceleration": ["x", "y"],
                    "heading": ["°", "d°"]},
    },
    "pred_state": {"VEHICLE": {"position": ["x", "y"]}},
    "fusion_hist_map_layer": 1,
    "mlp_layer": 2,
    "max_lane_num": 3,
    "sample_num": 10,
    "transformer": {"nlayers": 6, "nhead": 8, "in_dim": 256, "fdim": 512, "output_size": 32},
    "lane_encoder": {"VEHICLE": {"nlayers": 1, "map_channels": 3, "embedding_size": 2,
                                 "output_size": 32, "kernel_size": 2, "strides": 2, "dropout": 0.5}},
    # argparse-side keys (reference Transformer/train.py:33-45)
    "incl_robot_node": False,
    "edge_encoding": False,
    "map_cnn_encoding": False,
    "map_vit_encoding": False,
    "lane_cnn_encoding": False,
    "autoregressive": True,
    "augment": False,
    "override_attention_radius": [],
}


def make_hyperparams(mode: str = "basic", history: int = 8, horizon: int = 12,
                     transformer: dict | None = None, autoregressive: bool = True) -> dict:
    """Return the hyperparams dict for one of the four README configurations.

    mode: "basic" | "map_cnn" | "map_vit" | "lane_cnn"  (mutually exclusive flags,
    reference ``mgcvae.py:95,140,157``).
    """
    if mode not in MODES:
        raise ValueError(f"mode must be one of {MODES}, got {mode!r}")
    hp = copy.deepcopy(_BASE)
    hp["minimum_history_length"] = history
    hp["maximum_history_length"] = history
    hp["prediction_horizon"] = horizon
    hp["autoregressive"] = autoregressive
    hp["map_cnn_encoding"] = mode == "map_cnn"
    hp["map_vit_encoding"] = mode == "map_vit"
    hp["lane_cnn_encoding"] = mode == "lane_cnn"
    if transformer:
        hp["transformer"].update(transformer)
    return hp


def mode_of(hyperparams: dict) -> str:
    if hyperparams.get("lane_cnn_encoding"):
        return "lane_cnn"
    if hyperparams.get("map_vit_encoding"):
        return "map_vit"
    if hyperparams.get("map_cnn_encoding"):
        return "map_cnn"
    return "basic"
