"""Rehearsal of bench.py's multi-rank code path on ONE GPU: a one-rank RCCL group (backend nccl), the barrier / MAX all-reduce
bracketing of the timed region, and the training step with its flat gradient all-reduce issued between the two HIP-graph replays.
The losses must equal those of a run without a process group (a one-rank sum changes nothing).  python tools/rehearse_rccl.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402
from multimodal_transformer_amd.training import Trainer  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def losses(reduce):
    hp, model, reg = build_model("basic", dev, 8, 12)
    hp["autoregressive"] = True
    tr = Trainer(model, 128, 9, 12, use_graph=True, always_reduce=reduce)
    out = []
    for i in range(6):
        b = make_batch(128, 8, 12, seed=100 + i)
        out.append(float(tr.step(torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev))))
    b = make_batch(128, 8, 12, seed=1)
    pred = model.predict(torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["x_st"]).to(dev), None, None, None, 12)[0]
    return out, pred


plain, pred0 = losses(False)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
assert float(tt) == 1.5
reduced, pred1 = losses(True)
dist.barrier()
torch.cuda.synchronize()
print("losses without a group:", plain)
print("losses with the RCCL all-reduce between the graph replays:", reduced)
assert plain == reduced and torch.equal(pred0, pred1)
print("rehearsal ok", flush=True)
os._exit(0)        # (no communicator teardown: it has been seen to abort() the process)
