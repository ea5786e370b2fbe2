"""Profiling workload: N Basic training steps at batch B (rocprofv3 --kernel-trace target), eager (no graph) so that every
kernel shows up by name.  python tools/train_only.py [B] [N] [dropout 0/1] [graph 0/1] [sweep ways] [grouped weight gradients 0/1] [weight-gradient K slices]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402
from multimodal_transformer_amd.training import Trainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
drop = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
graph = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
dev = torch.device("cuda:0")
hp, model, reg = build_model("basic", dev, 8, 12)
hp["autoregressive"] = True
b = make_batch(B, 8, 12, seed=1)
x_st, y_st = torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev)
tr = Trainer(model, B, 9, 12, use_graph=graph, dropout=None if drop else False)
if len(sys.argv) > 5:
    tr.engine.sweep_ways = int(sys.argv[5])
if len(sys.argv) > 6:
    tr.engine.group_wgrads = bool(int(sys.argv[6]))
if len(sys.argv) > 7:
    tr.engine.SPLITK = int(sys.argv[7])          # K slices of the weight-gradient GEMMs (1: no mirrors to sum)
for _ in range(3):
    tr.step(x_st, y_st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    loss = tr.step(x_st, y_st)
torch.cuda.synchronize()
print("done", N, "steps", (time.perf_counter() - t0) / N * 1e3, "ms/step; loss", float(loss))
