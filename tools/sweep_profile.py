"""Diagnostic: per-visit wall-clock stamps (100 MHz) of the backward sweep pipeline (needs the -DMMT_PROFILE build:
make -C multimodal_transformer_amd/csrc prof; MMT_AMD_LIB=multimodal_transformer_amd/libmmt_amd_prof.so python tools/sweep_profile.py [B] [dropout])."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from multimodal_transformer_amd import _lib  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402
from multimodal_transformer_amd.training import TrainEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
drop = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = torch.device("cuda:0")
hp, model, reg = build_model("basic", dev, 8, 12)
hp["autoregressive"] = True
b = make_batch(B, 8, 12, seed=1)
eng = TrainEngine(model, B, 9, 12, dropout=None if drop else False)
eng.set_batch(torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev))
for _ in range(3):
    eng.forward()
    eng.backward()
torch.cuda.synchronize()
assert eng._sweep_pipe() is not None, eng.sweep_pipe_why
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
print(f"B={B} NC={NC} R={R} NHW={NHW} grid={R * per_rep}; total span {us(st.max() - t0):.1f} us")


def role(wg):
    rep, rb = divmod(wg, per_rep)
    if rb >= L * per_layer:
        return rep, "H", 0, rb - L * per_layer
    layer, q = divmod(rb, per_layer)           # mmt_bwd_pipe4_kernel: H self-attention, H cross-attention, H/2 feed-forward workgroups
    if q < H:
        return rep, "A", layer, q
    if q < 2 * H:
        return rep, "B", layer, q - H
    return rep, "C", layer, q - 2 * H


ncl = (NC + R - 1) // R
names = ["poll", "LN+stage(+bar)", "gemm1(+bar)", "attn(+kv units)", "gemm2", "end bar"]
for kind in "HCBA":
    seg = {k: [] for k in names}
    busy = []
    for wg in range(R * per_rep):
        rep, kd, layer, g = role(wg)
        if kd != kind or rep != 0:
            continue
        for s in range(2, 11):
            nl = ncl if kind != "H" else (ncl + NHW - 1) // NHW
            for li in range(nl):
                v = st[wg, s * 16 + li]
                if v[6] == 0:
                    continue
                for i, k in enumerate(names):
                    seg[k].append(us(v[i + 1] - v[i]))
                if kind in "AB":
                    seg.setdefault("attn_only", []).append(us(v[8] - v[3]))
                if kind != "H":
                    seg.setdefault("LN_only", []).append(us(v[7] - v[1]))
                    seg.setdefault("first_attempt", []).append(us(v[9] - v[0]))
                    seg.setdefault("full_fetches", []).append(float(v[10]) + 1.0)
                busy.append(us(v[6] - v[1]))
    print(kind, "visits", len(busy), " ".join(f"{k}={np.mean(v):.2f}" for k, v in seg.items() if v), f"busy(poll_end->end)={np.mean(busy):.2f}")

print("cluster 0 (replica 0), sweep step 3: stage  start_wait  poll_end ... (us since the step's head started)")
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
prev = None
for _, kd, layer, v in rows:
    hop = "" if prev is None else f"  hop {us(v[1] - prev):5.2f}"
    prev = v[1]
    print(f"  {kd}{layer}: start_wait {us(v[0]-base):7.2f} poll_end {us(v[1]-base):7.2f} staged {us(v[2]-base):7.2f} g1 {us(v[3]-base):7.2f} att {us(v[4]-base):7.2f} g2 {us(v[5]-base):7.2f} end {us(v[6]-base):7.2f}{hop}")

# hand-off anatomy: for every consumer visit (slice 0 of a stage), the store-issue stamps (5) of ALL producers of its input
stage_wgs = {}
for wg in range(per_rep):
    rep, kd, layer, g = role(wg)
    stage_wgs.setdefault((kd, layer), []).append(wg)
order = [("H", 0)] + [(k, l) for l in range(L - 1, -1, -1) for k in "CBA"]
for ci in range(1, len(order)):
    cons, prod = order[ci], order[ci - 1]
    if cons[1] not in (5, 2):
        continue
    tr, skew, qd, first = [], [], [], []
    cw = stage_wgs[cons][0]
    for s in range(2, 11):
        for li in range(ncl):
            v = st[cw, s * 16 + li]
            if v[6] == 0:
                continue
            if prod[0] == "H":
                pw = [w for w in stage_wgs[prod] if st[w, s * 16 + li // NHW][5] > 0 and (li % NHW) == (w - stage_wgs[prod][0])]
                g2 = [st[w, s * 16 + li // NHW][5] for w in pw]
            else:
                g2 = [st[w, s * 16 + li][5] for w in stage_wgs[prod]]
            if not g2 or min(g2) == 0:
                continue
            tr.append(us(v[1] - max(g2)))
            skew.append(us(max(g2) - min(g2)))
            first.append(us(max(g2) - g2[0]))
            qd.append(us(v[0] - max(g2)))
    if tr:
        print(f"{prod[0]}{prod[1]} -> {cons[0]}{cons[1]}: last producer's store issue -> consumer poll_end {np.mean(tr):.2f} us (p10 {np.percentile(tr, 10):.2f}, p90 {np.percentile(tr, 90):.2f}); "
              f"producer skew {np.mean(skew):.2f}; last minus slice 0: {np.mean(first):.2f}; consumer started waiting {np.mean(qd):.2f} us after the last store (negative: was waiting)")
