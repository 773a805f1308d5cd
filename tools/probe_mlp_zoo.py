"""Scratch: GGN matvec time at 8 rows for a few MLP shapes (no perf cliffs outside C2?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0")
def mlp(dims, act=nn.ReLU, bias=True):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1], bias=bias))
        if i < len(dims) - 2: layers.append(act())
    return nn.Sequential(*layers).to(dev)
zoo = {"C2 1024-2688-2688-10": [1024, 2688, 2688, 10], "deep 6x2048": [2048] * 7 + [10], "wide head 2688-2688-1000": [1024, 2688, 2688, 1000],
       "narrow 256x4": [256, 256, 256, 256, 10], "single layer 4096->4096": [4096, 4096], "odd widths 1001-2051-77": [1001, 2051, 77],
       "huge 8192-8192-8192-10": [8192, 8192, 8192, 10]}
for name, dims in zoo.items():
    torch.manual_seed(0)
    model = mlp(dims); params = dict(model.named_parameters())
    for N in (8, 128):
        X, y = torch.rand(N, dims[0], device=dev), torch.rand(N, dims[-1], device=dev)
        G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
        D = G.shape[1]; vs = [torch.rand(D, device=dev) for _ in range(4)]
        for i in range(3): G @ vs[i]
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 40
        for i in range(n): G @ vs[i % 4]
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
        print(f"{name:28s} D={D/1e6:6.1f}M N={N:3d}: {t*1e6:8.1f} us  {12*D/t/1e12:.2f} TB/s alg  {10*N*D/t/1e12:.1f} TF alg")
