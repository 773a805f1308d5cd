"""Exact-Hessian K columns on C2 (clo_mlp_hessian_matmat) against K single-vector products."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
params = dict(model.named_parameters())
X, y = torch.rand(8, 1024, device=dev), torch.rand(8, 10, device=dev)
H = C.HessianLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
D = H.shape[1]
def t_us(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
v = torch.rand(D, device=dev)
print(f"single vector: {t_us(lambda: H @ v, 50):.1f} us", flush=True)
for K in (8, 16, 32, 64):
    V = torch.rand(D, K, device=dev)
    t = t_us(lambda: H @ V)
    a = H @ V
    b = H @ V[:, K - 1].contiguous()
    err = float((a[:, K - 1] - b).abs().max() / b.abs().max())
    print(f"K = {K}: {t:.0f} us = {t / K:.1f} us per column ({20 * D / (t / K) / 1e6:.2f} TB/s on 20 D B), last column vs matvec {err:.1e}", flush=True)
