"""Scratch: damped-GGN solve on C2 (D = 10M) by CG on the native matvec: time per iteration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
params = dict(model.named_parameters())
X, y = torch.rand(8, 1024, device=dev), torch.rand(8, 10, device=dev)
G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
A = G + C.DiagonalLinearOperator.identity_like(G, 1e-3)
b = torch.rand(G.shape[1], device=dev)
for iters in (20, 100):
    inv = C.CGInverseLinearOperator(A, max_iter=iters, tolerance=0.0)
    inv @ b; torch.cuda.synchronize(); t0 = time.perf_counter()
    x = inv @ b; torch.cuda.synchronize(); t = time.perf_counter() - t0
    res = (A @ x - b).norm() / b.norm()
    print(f"CG {iters} iterations on D={G.shape[1]}: {t*1e3:.2f} ms = {t/iters*1e6:.0f} us/iteration, rel residual {res:.2e}")
