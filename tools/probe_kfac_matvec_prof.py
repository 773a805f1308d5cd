"""Kernel composition of one KFAC matvec on ResNet-18 (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18(num_classes=10).to(dev).eval()
params = {n: p for n, p in model.named_parameters() if p.dim() in (2, 4) or "bias" in n and ("conv" in n or "fc" in n or "linear" in n)}
params = {n: p for n, p in model.named_parameters() if any(n.startswith(m) for m, mod in model.named_modules() if isinstance(mod, (nn.Conv2d, nn.Linear)))}
X = torch.rand(128, 3, 32, 32, device=dev); y = torch.randint(0, 10, (128,), device=dev)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="empirical",
                         separate_weight_and_bias=False, check_deterministic=False)
v = torch.rand(K.shape[1], device=dev)
for _ in range(3): K @ v
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): K @ v
torch.cuda.synchronize(); print(f"kfac matvec {(time.perf_counter()-t)/20*1e3:.3f} ms", flush=True)
