"""Time of the training step's autoregressive decode with and without the stash (HIP events around TrainEngine._ar_decode).
python tools/time_stash_decode.py [B] [dropout 0/1]      (MMT_AMD_LIB selects a library build)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import build_model
from multimodal_transformer_amd.synthetic import make_batch
from multimodal_transformer_amd.training import TrainEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
drop = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = torch.device("cuda:0")
hp, model, reg = build_model("basic", dev, 8, 12)
hp["autoregressive"] = True
b = make_batch(B, 8, 12, seed=1)
eng = TrainEngine(model, B, 9, 12, dropout=None if drop else False)
eng.set_batch(torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev))
eng.forward()
torch.cuda.synchronize()
for stash in (False, True, False, True):
    for _ in range(10):
        eng._ar_decode(stash=stash)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        eng._ar_decode(stash=stash)
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('MMT_AMD_LIB', 'default')} stash={stash}: {e0.elapsed_time(e1) / 100:.4f} ms per decode (incl. re-layout + ckv + final head)", flush=True)
