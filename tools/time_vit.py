"""Diagnostic: the ViT map encoder forward as the fused kernel (mmt_vit_forward) vs as batched GEMMs (training._VitTape.forward):
python tools/time_vit.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from multimodal_transformer_amd import training
from multimodal_transformer_amd.synthetic import make_batch
dev = torch.device("cuda:0")
for B in (128, 1024):
    hp, model, reg = build_model("map_vit", dev, 8, 12)
    b = make_batch(B, 8, 12, seed=1, with_map=True)
    maps = torch.from_numpy(b["map"]).to(dev)
    vit = reg.model_dict["VEHICLE/map_encoder"]
    class _NoDrop:
        view = {}
    class E:
        device = dev
        drop = _NoDrop()
        def _buf(self, *shape, dtype=torch.float32): return torch.zeros(*shape, dtype=dtype, device=dev)
    tape = training._VitTape(E(), vit, B, 2.0, -1.0)
    def avg(fn, n=30):
        fn(); torch.cuda.synchronize()
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        c.record(); torch.cuda.synchronize()
        return a.elapsed_time(c) / n
    y1 = vit.forward_raw(maps); y2 = tape.forward(maps)
    print(B, "fused kernel ms", avg(lambda: vit.forward_raw(maps)), "gemm tape ms", avg(lambda: tape.forward(maps)), "maxdiff", float((y1 - y2).abs().max()))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tape.forward(maps); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            tape.forward(maps)
    print(B, "gemm tape graph ms", avg(lambda: g.replay()))
