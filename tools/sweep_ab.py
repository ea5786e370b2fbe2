"""Loss trajectory of the training step with the pipeline sweep against the launch-per-kernel sweep (graph replay), same batch.
python tools/sweep_ab.py [steps] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402
from multimodal_transformer_amd.training import Trainer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
out = []
for pipe in (True, False):
    hp, model, reg = build_model("basic", dev, 8, 12)
    hp["autoregressive"] = True
    tr = Trainer(model, B, 9, 12, use_graph=True, dropout=False)
    tr.engine.sweep_pipeline = pipe
    losses, gn = [], []
    for it in range(N):
        b = make_batch(B, 8, 12, seed=1 + it % 4)
        losses.append(float(tr.step(torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev))))
        gn.append(float(tr.engine.flat_grad[:tr.engine.guard_at].double().norm()))
    out.append((losses, gn, tr.engine.sweep_pipe_why if pipe else "off"))
for i in range(N):
    print(i, "loss %.7f %.7f   |grad| %.6e %.6e" % (out[0][0][i], out[1][0][i], out[0][1][i], out[1][1][i]))
print(out[0][2], "| skipped", )
