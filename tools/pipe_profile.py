"""Diagnostic: per-visit wall-clock stamps (100 MHz) of the weight-stationary pipeline decoder (needs the -DMMT_PROFILE build:
make -C multimodal_transformer_amd/csrc prof; MMT_AMD_LIB=multimodal_transformer_amd/libmmt_amd_prof.so python tools/pipe_profile.py [B])."""
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
ops.decoder_select("cluster_pipeline")
hp, model, reg = build_model("basic", dev, 8, 12)
b = make_batch(B, 8, 12, seed=1)
x, x_st = torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["x_st"]).to(dev)
for _ in range(3):
    model.predict(x, x_st, None, None, None, 12)
torch.cuda.synchronize()
lib = _lib.load()
lib.mmt_debug_read_pprof.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
n = 256 * 192 * 16
buf = (ctypes.c_ulonglong * n)()
assert lib.mmt_debug_read_pprof(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 192, 16).astype(np.int64)
L, H = 6, 8
per_layer = 2 * H + H // 2
NC = (B + 3) // 4
R = min(NC, 256 // (L * per_layer + 1))
NHW = max(1, min(min((256 - R * L * per_layer) // R, 4), (NC + R - 1) // R))
per_rep = L * per_layer + NHW
t0 = st[st > 0].min()
us = lambda v: v / 100.0
names = ["kv+poll", "LN", "gemm1", "attn", "gemm2", "end"]
print(f"B={B} NC={NC} R={R} NHW={NHW} grid={R * per_rep}; total span {us(st.max() - t0):.1f} us")


def role(wg):
    rep, rb = divmod(wg, per_rep)
    if rb >= L * per_layer:
        return rep, "H", 0, rb - L * per_layer
    layer, q = divmod(rb, per_layer)
    if q < H:
        return rep, "A", layer, q
    if q < H + H // 2:
        return rep, "B", layer, q - H
    return rep, "C", layer, q - H - H // 2


ncl = (NC + R - 1) // R
for kind in "HABC":
    seg = {k: [] for k in names}
    busy = []
    for wg in range(R * per_rep):
        rep, kd, layer, g = role(wg)
        if kd != kind or rep != 0:
            continue
        for s in range(2, 12):
            nl = ncl if kind != "H" else (ncl + NHW - 1) // NHW
            for li in range(nl):
                v = st[wg, s * 16 + li]
                if v[6] == 0:
                    continue
                for i, k in enumerate(names):
                    a, bnd = v[i], v[i + 1]
                    if kind == "H" and i in (2, 3):      # head: stamps 2, 3 unused
                        continue
                    if kind == "H" and i == 1:
                        bnd = v[4]
                    if kind == "C" and i == 3:
                        pass
                    seg[k].append(us(bnd - a))
                if kind in "AB":
                    seg.setdefault("g1_only", []).append(us(v[7] - v[2]))
                    seg.setdefault("qk", []).append(us(v[8] - v[3]))
                    seg.setdefault("softmax", []).append(us(v[9] - v[8]))
                    seg.setdefault("pv", []).append(us(v[10] - v[9]))
                    seg.setdefault("att_bar", []).append(us(v[4] - v[10]))
                if kind != "H":
                    seg.setdefault("LN_only", []).append(us(v[11] - v[1]))
                busy.append(us(v[6] - v[1]))
    print(kind, "visits", len(busy), " ".join(f"{k}={np.mean(v):.2f}" for k, v in seg.items() if v), f"busy(wait_end->arrive)={np.mean(busy):.2f}")

# timeline of cluster 0 (replica 0, local index 0) through step 3
print("cluster 0, step 3: stage  wait_end  arrive (us since the step's head started)")
s = 3
rows = []
for wg in range(per_rep):
    rep, kd, layer, g = role(wg)
    if g != 0:
        continue
    v = st[wg, s * 16 + 0]
    rows.append((v[1], kd, layer, v))
rows.sort(key=lambda r: r[0])
base = rows[0][3][0]
for _, kd, layer, v in rows:
    print(f"  {kd}{layer}: start_wait {us(v[0]-base):7.2f} wait_end {us(v[1]-base):7.2f} loaded {us(v[2]-base):7.2f} g1 {us(v[3]-base):7.2f} att {us(v[4]-base):7.2f} g2 {us(v[5]-base):7.2f} arrived {us(v[6]-base):7.2f}")

# shader clock (s_memtime) against the 100 MHz wall clock over one visit's busy part: the frequency the chip holds in this kernel
fr = []
for wg in range(R * per_rep):
    if role(wg)[1] not in "ABC":
        continue
    for s in range(2, 12):
        for li in range(ncl):
            v = st[wg, s * 16 + li]
            if v[6] and v[12] and v[13]:
                fr.append((v[13] - v[12]) / max(1, (v[6] - v[0])) * 100.0)
if fr:
    print(f"shader clock over a visit: median {np.median(fr):.0f} MHz (p10 {np.percentile(fr, 10):.0f}, p90 {np.percentile(fr, 90):.0f})")
