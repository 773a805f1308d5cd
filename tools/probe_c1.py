"""Scratch: BASELINE config C1 (MLP 128-256-64-10 Tanh, MSE mean, 2 batches of 64) on the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = nn.Sequential(nn.Linear(128, 256), nn.Tanh(), nn.Linear(256, 64), nn.Tanh(), nn.Linear(64, 10)).to(dev)
params = dict(model.named_parameters())
data = [(torch.rand(64, 128, device=dev), torch.rand(64, 10, device=dev)) for _ in range(2)]
for name, cls in (("ggn", C.GGNLinearOperator), ("hessian", C.HessianLinearOperator), ("ef", C.EFLinearOperator)):
    for native in (True, False):
        op = cls(model, nn.MSELoss(), params, data, check_deterministic=False)
        if not native: op._native = None
        v = torch.rand(op.shape[1], device=dev)
        for _ in range(5): op @ v
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        for _ in range(n): op @ v
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
        print(f"C1 {name:8s} {'native  ' if native else 'autograd'}: {t*1e6:8.0f} us per matvec")
