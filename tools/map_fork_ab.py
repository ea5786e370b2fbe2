"""Map encoder on a second stream beside the history encoder, or in line before it (the split encoder's 256 spin-waiting
workgroups and the map kernels compete for CUs).  python tools/map_fork_ab.py"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from bench import build_model, predict_fn
dev = torch.device("cuda:0")
for mode in ("map_cnn", "map_vit"):
    hp, model, reg = build_model(mode, dev, 8, 12)
    step, tensors, batch = predict_fn(model, mode, dev, 128, 8, 12, seed=1)
    for graphed in (False, True):
        for fork in (True, False, True, False):
            model.fork_map_encoder = fork
            model.graph_predict = graphed
            if graphed:
                model.reset_predict_graphs()
            for _ in range(20): step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300): step()
            torch.cuda.synchronize()
            print(f"{mode} graph={graphed} fork={fork}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per predict", flush=True)
