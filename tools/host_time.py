import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from bench import build_model, predict_fn
dev = torch.device("cuda:0")
hp, model, reg = build_model("basic", dev, 8, 12)
step, tensors, batch = predict_fn(model, "basic", dev, 128, 8, 12, seed=1)
for _ in range(20): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue per predict %.3f ms; total per predict %.3f ms" % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
