"""Scratch: C1 native GGN / Hessian only (for rocprof kernel lists)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = nn.Sequential(nn.Linear(128, 256), nn.Tanh(), nn.Linear(256, 64), nn.Tanh(), nn.Linear(64, 10)).to(dev)
params = dict(model.named_parameters())
data = [(torch.rand(64, 128, device=dev), torch.rand(64, 10, device=dev)) for _ in range(2)]
kind = sys.argv[1] if len(sys.argv) > 1 else "ggn"
cls = {"ggn": C.GGNLinearOperator, "hessian": C.HessianLinearOperator}[kind]
op = cls(model, nn.MSELoss(), params, data, check_deterministic=False)
v = torch.rand(op.shape[1], device=dev)
for _ in range(5): op @ v
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 100
for _ in range(n): op @ v
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
print(f"C1 {kind}: {t*1e6:.0f} us per matvec")
