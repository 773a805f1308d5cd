"""Scratch: is the public-API matvec loop host-bound?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
params = dict(model.named_parameters())
X, y = torch.rand(8, 1024, device=dev), torch.rand(8, 10, device=dev)
G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
D = G.shape[1]
vs = [torch.rand(D, device=dev) for _ in range(8)]
for i in range(20): G @ vs[i % 8]
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for i in range(n): out = G @ vs[i % 8]
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host loop {1e6*(t1-t0)/n:.1f} us/matvec, total {1e6*(t2-t0)/n:.1f} us/matvec")
# raw plan call
nat = G._native
V = [torch.rand_like(p) for p in params.values()]; O = [torch.empty_like(p) for p in params.values()]
for i in range(20): nat.matvec(V, O, X, 0, 2.0/80, 1.0, 0.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n): nat.matvec(V, O, X, 0, 2.0/80, 1.0, 0.0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"raw plan: host loop {1e6*(t1-t0)/n:.1f} us, total {1e6*(t2-t0)/n:.1f} us")
