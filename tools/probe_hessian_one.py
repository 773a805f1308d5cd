"""Scratch: C2 Hessian matvec at N rows, native only (for rocprof kernel chains)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0"); torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
params = dict(model.named_parameters())
X, y = torch.rand(N, 1024, device=dev), torch.rand(N, 10, device=dev)
H = C.HessianLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
vs = [torch.rand(H.shape[1], device=dev) for _ in range(4)]
for i in range(3): H @ vs[i]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): H @ vs[i % 4]
torch.cuda.synchronize(); print(f"N={N}: {(time.perf_counter()-t0)/100*1e6:.0f} us")
