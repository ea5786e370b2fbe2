"""A/B of one LayerNorm forward / backward launch (MMT_TRAIN_GENERIC picks the form): python tools/ab_ln_forms.py out.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_transformer_amd import _train_kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N = 1536, 256
x, gam, bet = torch.randn(M, N, generator=g).to(dev), torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
y, xh, rs = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), torch.empty(M, device=dev)
K.layernorm_fwd(x, None, gam, bet, y, xh, rs)
dy, dx = torch.randn(M, N, generator=g).to(dev), torch.empty(M, N, device=dev)
K.layernorm_bwd(dy, xh, rs, gam, dx)
torch.cuda.synchronize()
torch.save({"y": y.cpu(), "xhat": xh.cpu(), "rstd": rs.cpu(), "dx": dx.cpu()}, sys.argv[1])
