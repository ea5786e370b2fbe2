"""Diagnostic: HIP-event time of the encoder and of one decode at batch B (MMT_AMD_LIB selects a library build):
python tools/time_decoder.py [B] [decoder]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, time_kernels  # noqa: E402
from multimodal_transformer_amd import ops  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
ops.decoder_select(sys.argv[2] if len(sys.argv) > 2 else "auto")
hp, model, reg = build_model("basic", dev, 8, 12)
b = make_batch(B, 8, 12, seed=1)
x, x_st = torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["x_st"]).to(dev)
out = model.predict(x, x_st, None, None, None, 12)[0]
print(os.environ.get("MMT_AMD_LIB", "default"), "B=%d" % B, time_kernels(model, reg, (x, x_st, None, None), 12, 100),
      "finite" if bool(torch.isfinite(out).all()) else "NOT FINITE", float(out.double().sum()))
