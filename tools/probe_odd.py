import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0")
dims = [1001, 2051, 77]
torch.manual_seed(0)
model = nn.Sequential(nn.Linear(1001, 2051), nn.ReLU(), nn.Linear(2051, 77)).to(dev); params = dict(model.named_parameters())
for rep in range(3):
    for N in (8, 128):
        X, y = torch.rand(N, 1001, device=dev), torch.rand(N, 77, device=dev)
        G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
        D = G.shape[1]; vs = [torch.rand(D, device=dev) for _ in range(4)]
        for i in range(3): G @ vs[i]
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 40
        for i in range(n): G @ vs[i % 4]
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
        print(f"odd N={N}: {t*1e6:.1f} us")
