"""Profiling workload: N Basic predicts at batch B with a chosen decoder form (rocprofv3 --kernel-trace / --pmc target).
python tools/decode_only.py [B] [decoder] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from multimodal_transformer_amd import ops  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 25
dev = torch.device("cuda:0")
ops.decoder_select(sys.argv[2] if len(sys.argv) > 2 else "auto")
hp, model, reg = build_model("basic", dev, 8, 12)
b = make_batch(B, 8, 12, seed=1)
x, x_st = torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["x_st"]).to(dev)
for _ in range(N):
    out = model.predict(x, x_st, None, None, None, 12)[0]
torch.cuda.synchronize()
ops.decoder_status()
print("done", N, "predicts; finite:", bool(torch.isfinite(out).all()))
