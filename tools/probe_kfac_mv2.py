"""Scratch: ResNet-18 KFAC matvec only (for rocprof kernel stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18().to(dev).eval(); params = kfac_params(model); B = 128
X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False,
                         check_deterministic=False, num_data=B)
v = torch.rand(K.shape[1], device=dev)
for _ in range(3): K @ v
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): r = K @ v
torch.cuda.synchronize(); print(f"{(time.perf_counter()-t0)/20*1e3:.3f} ms per KFAC matvec")
