"""Diagnostic: per-phase shader-cycle stamps of one decoder layer-step (needs the -DMMT_PROFILE build:
MMT_AMD_LIB=multimodal_transformer_amd/libmmt_amd_prof.so python tools/phase_profile.py [B])."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from multimodal_transformer_amd import _lib  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
hp, model, reg = build_model("basic", dev, 8, 12)
b = make_batch(B, 8, 12, seed=1)
x, x_st = torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["x_st"]).to(dev)
for _ in range(3):
    model.predict(x, x_st, None, None, None, 12)
torch.cuda.synchronize()
lib = _lib.load()
lib.mmt_debug_read_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 512)()
assert lib.mmt_debug_read_prof(buf, 512) == 0
names = ["qkv", "kv-store", "self-attn", "out-proj", "ln1", "cross-q", "cross-attn", "cross-out", "ln2", "ff1", "ff2",
         "ln3"]
for L in range(6):
    st = [buf[L * 16 + k] for k in range(13)]
    d = [st[i + 1] - st[i] for i in range(12)]
    print(f"layer {L}: total {st[12] - st[0]:7d} cyc | " + " ".join(f"{n}={v}" for n, v in zip(names, d)))

# history encoder (workgroup 0): wall-clock (100 MHz) stamps per layer
names = ["qkv", "attention", "out-proj", "ln1", "ff1", "ff2", "ln2"]
for L in range(6):
    st = [buf[256 + L * 8 + k] for k in range(8)]
    d = [(st[i + 1] - st[i]) / 100.0 for i in range(7)]
    print(f"encoder layer {L}: total {(st[7] - st[0]) / 100.0:6.2f} us | " + " ".join(f"{n}={v:.2f}" for n, v in zip(names, d)))
