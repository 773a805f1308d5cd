"""Scratch: C2 GGN `A @ M` with K columns (operator API, [D, K] K-trailing) vs K single matvecs."""
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
v = torch.rand(D, device=dev)
for _ in range(5): G @ v
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): G @ v
torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 50
print(f"K=1: {t1*1e6:.1f} us")
for K in [int(a) for a in sys.argv[1:]] or [4, 8, 32, 64]:
    Vs = [torch.rand(D, K, device=dev) for _ in range(2)]
    for i in range(2): G @ Vs[i]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 6
    for i in range(n): out = G @ Vs[i % 2]
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print(f"K={K}: {t*1e6:.1f} us per matmat = {t/K*1e6:.1f} us per column ({t1*K/t:.2f}x vs K matvecs); "
          f"streamed {8*D*K/t/1e12:.2f} TB/s (V read + result written)")
