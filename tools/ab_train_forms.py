"""One full-size Basic training step (forward + backward), every stash tensor and the gradient saved: python tools/ab_train_forms.py
out.pt [dropout 0/1].  Run under MMT_TRAIN_GENERIC=0 / 7 to compare the register-resident LayerNorm / burst attention kernels with the
generic forms (tests/test_gpu_training.py does)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from multimodal_transformer_amd.synthetic import make_batch
from multimodal_transformer_amd.training import TrainEngine
dev = torch.device("cuda:0")
hp, model, reg = build_model("basic", dev, 8, 12)
hp["autoregressive"] = True
b = make_batch(128, 8, 12, seed=1)
eng = TrainEngine(model, 128, 9, 12, dropout=bool(int(sys.argv[2])) if len(sys.argv) > 2 else False, seed=3)
eng.set_batch(torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev))
loss = float(eng.forward()); eng.backward(); torch.cuda.synchronize()
S = eng.s
out = {"loss": torch.tensor(loss), "grad": eng.flat_grad.cpu()}
for n in ("eh", "eh1", "exhat1", "eatt", "x", "x1", "x2", "a1", "a2", "xhat1", "xhat3", "gx", "gx1", "geh", "gef2"):
    for i, t in enumerate(getattr(S, n)): out[f"{n}{i}"] = t.cpu()
torch.save(out, sys.argv[1])
print("loss", loss)
