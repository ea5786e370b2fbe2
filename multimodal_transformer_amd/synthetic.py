"""nuScenes-shaped synthetic batches.

No dataset can be downloaded, so the bench, the tests and the golden generator
all use this deterministic generator.  Shapes and units follow what the
reference's pre-processing hands to ``Trajectron.train_loss/predict``
(reference ``Transformer/model/dataset/preprocessing.py:100-118`` for x / y /
x_st / y_st; ``Transformer/environment/environment.py:61`` for the divide-by-80
standardisation; ``Transformer/environment/data_utils.py:20-30`` for the
np.gradient derivatives).
"""
from __future__ import annotations

import numpy as np

DT = 0.5            # nuScenes sample period used by the reference (2 Hz)
ATTENTION_RADIUS = 80.0   # every state dim is divided by this (environment.py:61)


def _box_blur(img: np.ndarray, k: int = 3) -> np.ndarray:
    """Cheap separable box blur on the last two axes (mimics bilinear-rotated rasters)."""
    pad = k // 2
    out = img
    for axis in (-1, -2):
        p = [(0, 0)] * out.ndim
        p[axis] = (pad, pad)
        padded = np.pad(out, p, mode="edge")
        acc = np.zeros_like(out)
        for s in range(k):
            sl = [slice(None)] * out.ndim
            sl[axis] = slice(s, s + out.shape[axis])
            acc = acc + padded[tuple(sl)]
        out = acc / k
    return out


def make_batch(batch: int, history: int = 8, horizon: int = 12, seed: int = 1,
               with_map: bool = False, with_lanes: bool = False, map_size: int = 100,
               lane_num: int = 3, sample_num: int = 10) -> dict:
    """Return a dict of float32 numpy arrays:

    x (B,T,8) metres, x_st (B,T,8), y (B,ph,2), y_st (B,ph,2) with T = history+1,
    optionally map (B,3,S,S) in [0,1] and lanes (B,lane_num,sample_num,2).
    State column order: [x, y, vx, vy, ax, ay, heading(rad), dheading].
    """
    rng = np.random.default_rng(seed)
    T = history + 1
    n = T + horizon
    pos0 = rng.uniform(50.0, 500.0, size=(batch, 2))
    speed = rng.uniform(0.0, 15.0, size=(batch,))
    heading0 = rng.uniform(-np.pi, np.pi, size=(batch,))
    yaw_rate = rng.normal(0.0, 0.1, size=(batch,))
    accel = rng.normal(0.0, 1.0, size=(batch,))

    t = np.arange(n) * DT
    heading = heading0[:, None] + yaw_rate[:, None] * t[None, :]
    v = np.clip(speed[:, None] + accel[:, None] * t[None, :], 0.0, None)
    vx = v * np.cos(heading)
    vy = v * np.sin(heading)
    px = pos0[:, 0:1] + np.cumsum(vx * DT, axis=1) - vx[:, 0:1] * DT
    py = pos0[:, 1:2] + np.cumsum(vy * DT, axis=1) - vy[:, 0:1] * DT
    # derivatives exactly as the reference's derivative_of(): np.gradient over time
    gvx = np.gradient(px, DT, axis=1)
    gvy = np.gradient(py, DT, axis=1)
    gax = np.gradient(gvx, DT, axis=1)
    gay = np.gradient(gvy, DT, axis=1)
    dh = np.gradient(heading, DT, axis=1)
    state = np.stack([px, py, gvx, gvy, gax, gay, heading, dh], axis=-1)

    x = state[:, :T, :]
    y = state[:, T:, 0:2]
    rel = np.zeros((batch, 1, 8))
    rel[:, 0, 0:2] = x[:, -1, 0:2]
    x_st = (x - rel) / ATTENTION_RADIUS
    y_st = (y - x[:, -1:, 0:2]) / ATTENTION_RADIUS

    out = {
        "x": x.astype(np.float32),
        "x_st": x_st.astype(np.float32),
        "y": y.astype(np.float32),
        "y_st": y_st.astype(np.float32),
    }
    if with_map:
        coarse = (rng.uniform(size=(batch, 3, map_size // 5, map_size // 5)) > 0.5).astype(np.float64)
        fine = np.repeat(np.repeat(coarse, 5, axis=-1), 5, axis=-2)
        out["map"] = _box_blur(fine, 3).astype(np.float32)
    if with_lanes:
        # lane sample points 0.1 .. 1 m ahead along (slightly perturbed) headings, /80;
        # an absent lane is filled with -100/80 = -1.25 (reference lane_utils.py:279-282)
        d = np.linspace(0.1, 1.0, sample_num)
        lanes = np.empty((batch, lane_num, sample_num, 2))
        for k in range(lane_num):
            hk = heading[:, T - 1] + rng.normal(0.0, 0.3, size=(batch,))
            lanes[:, k, :, 0] = d[None, :] * np.cos(hk)[:, None] / ATTENTION_RADIUS
            lanes[:, k, :, 1] = d[None, :] * np.sin(hk)[:, None] / ATTENTION_RADIUS
        absent = rng.uniform(size=(batch, lane_num)) < 0.15
        absent[:, 0] = False
        lanes[absent] = -1.25
        out["lanes"] = lanes.astype(np.float32)
    return out
