"""Host time of a predict: enqueue time per call (eager and graph-replayed), wall time of a synchronous caller (predict + .cpu()),
and a cProfile of the eager path.  python tools/host_time.py [mode]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from bench import build_model, predict_fn, sync_predict_ms
mode = sys.argv[1] if len(sys.argv) > 1 else "basic"
dev = torch.device("cuda:0")
hp, model, reg = build_model(mode, dev, 8, 12)
step, tensors, batch = predict_fn(model, mode, dev, 128, 8, 12, seed=1)
for graphed in (False, True):
    model.graph_predict = graphed
    for _ in range(20): step()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s: host enqueue per predict %.3f ms; total per predict %.3f ms" % ("graph" if graphed else "eager", (t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3), flush=True)
model.graph_predict = False
print("synchronous caller (predict + .cpu()), ms per call:", sync_predict_ms(model, mode, dev, 128, 8, 12), flush=True)
import cProfile, pstats
for graphed in (True, False):
    model.graph_predict = graphed
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): step()
    pr.disable(); torch.cuda.synchronize()
    print("---- cProfile,", "graph" if graphed else "eager")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
