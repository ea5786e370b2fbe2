"""Diagnostic: wall-clock stamps of the cluster decoder's launches (needs the -DMMT_PROFILE build:
make -C multimodal_transformer_amd/csrc prof; MMT_AMD_LIB=multimodal_transformer_amd/libmmt_amd_prof.so python tools/cluster_profile.py [B])."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from multimodal_transformer_amd import _lib, ops  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
ops.decoder_select("cluster")
hp, model, reg = build_model("basic", dev, 8, 12)
b = make_batch(B, 8, 12, seed=1)
x, x_st = torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["x_st"]).to(dev)
for _ in range(3):
    model.predict(x, x_st, None, None, None, 12)
torch.cuda.synchronize()
lib = _lib.load()
lib.mmt_debug_read_clprof.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
n = 256 * 64 * 8
buf = (ctypes.c_ulonglong * n)()
assert lib.mmt_debug_read_clprof(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 64, 8).astype(np.int64)
rc = int(os.environ.get('MMT_CLUSTER_ROWS', '4' if B <= 128 else '16'))
nwg = min(64, ((B + rc - 1) // rc) * 8)
names = ["issue", "wait+LN(+head)", "gemm1", "attn", "gemm2"]
print("launch kind | first-WG start after previous launch's last end (us) | start skew | per-stage mean over WGs (us) | total")
prev_end = None
for p in range(60, 60 + 3 * 8):          # a few layers in the middle of the decode
    w = st[p, :nwg]
    start, end = w[:, 0], w[:, 5]
    gap = (start.min() - prev_end) / 100.0 if prev_end is not None else float("nan")
    stages = [(w[:, k + 1] - w[:, k]).mean() / 100.0 for k in range(5)]
    print(f"{p:4d} {'ABC'[p % 3]} gap={gap:6.2f} skew={(start.max() - start.min()) / 100.0:5.2f} " +
          " ".join(f"{nm}={v:5.2f}" for nm, v in zip(names, stages)) + f" | span={(end.max() - start.min()) / 100.0:6.2f}")
    prev_end = end.max()


if st[61, 0, 6] > 0:      # the one-launch form stamps the arrival too
    print("fused kernel: per-phase stage means over the first %d workgroups (us)" % nwg)
    names = ["wait+issue", "LN(+head)", "gemm1", "attn", "gemm2", "drain+arrive"]
    tot = np.zeros(7)
    for p in range(36, 36 + 18):
        w = st[p, :nwg]
        stages = [(w[:, k + 1] - w[:, k]).mean() / 100.0 for k in range(6)]
        nxt = (st[p + 1, :nwg, 0] - w[:, 6]).mean() / 100.0
        tot += np.array(stages + [nxt])
        print(f"{p:4d} {'ABC'[p % 3]} " + " ".join(f"{nm}={v:5.2f}" for nm, v in zip(names, stages)) + f" next={nxt:5.2f} | phase={(st[p + 1, :nwg, 0] - w[:, 0]).mean() / 100.0:5.2f}")
    print("mean per phase:", " ".join(f"{nm}={v / 18:5.2f}" for nm, v in zip(names + ["next"], tot)), f"sum={tot.sum() / 18:5.2f}")
