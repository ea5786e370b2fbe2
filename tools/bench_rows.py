import sys, torch
sys.path.insert(0, '/root/repo')
from multimodal_transformer_amd import _train_kernels as K
dev = torch.device('cuda:0')
def avg(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a.record(); g.replay(); c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / n * 1e3
for (M, Kd, N) in ((128, 768, 256), (128, 512, 256), (128, 256, 256), (128, 256, 512)):
    A = torch.randn(M, Kd, device=dev); W = torch.randn(Kd, N, device=dev); Wt = W.t().contiguous()
    C = torch.empty(M, N, device=dev)
    t_nat = avg(lambda: K.gemm(A, W, C))
    t_tr = avg(lambda: K.gemm(A, Wt.t(), C))
    print(M, Kd, N, "B n-contiguous %.2f us   B k-contiguous %.2f us" % (t_nat, t_tr))
