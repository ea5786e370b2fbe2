"""Diagnostic: predict rate with the inputs starting in HOST memory (pinned) instead of HBM - the PCIe-inclusive figure DESIGN.md
quotes beside the bench's HBM-resident one.  python tools/time_pcie.py [mode] [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "basic"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
hp, model, reg = build_model(mode, dev, 8, 12)
b = make_batch(B, 8, 12, seed=1, with_map=mode in ("map_cnn", "map_vit"), with_lanes=mode == "lane_cnn")
host = {k: torch.from_numpy(v).pin_memory() for k, v in b.items() if k in ("x", "x_st", "lanes", "map")}
nbytes = sum(t.numel() * t.element_size() for t in host.values())


def resident(t):
    return model.predict(t["x"], t["x_st"], t.get("lanes"), None, t.get("map"), 12)


def from_host():
    t = {k: v.to(dev, non_blocking=True) for k, v in host.items()}
    out = resident(t)
    return (out[1] if mode == "lane_cnn" else out[0]).to("cpu", non_blocking=True)


t_dev = {k: v.to(dev) for k, v in host.items()}
for name, f in (("inputs in HBM", lambda: resident(t_dev)), ("inputs in pinned host memory, result copied back", from_host)):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{mode} B={B} {name}: {dt * 1e3:.3f} ms/predict, {B / dt:,.0f} agents/s ({nbytes / 1e6:.2f} MB of inputs)")
