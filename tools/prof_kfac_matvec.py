"""KFAC / EKFAC matvec on ResNet-18 (joint W+b factors: odd orders) -- which GEMM kernels serve it?  Run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18().to(dev).eval(); params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
kw = dict(fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=512)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
v = torch.rand(K.shape[1], device=dev)
for _ in range(3): K @ v
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): K @ v
torch.cuda.synchronize(); print(f"kfac matvec {1e3 * (time.perf_counter() - t0) / 20:.3f} ms", flush=True)
from curvlinops_amd import _hip
mark = torch.zeros(4099, device=dev)
torch.cuda.synchronize()
_hip.axpby(mark, mark, 1.0, 0.0)   # marker launches around ONE warm product (tools/kfac_trace_summary.py)
K @ v
_hip.axpby(mark, mark, 1.0, 0.0)
torch.cuda.synchronize()
V8 = torch.rand(K.shape[1], 8, device=dev)
for _ in range(3): K @ V8
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): K @ V8
torch.cuda.synchronize(); print(f"kfac matmat K=8 {1e3 * (time.perf_counter() - t0) / 10:.3f} ms", flush=True)
