"""Scratch: three ResNet-18 KFAC factor builds (B = 512, joint W+b, mc) separated by sleeps, for timeline profiling."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
kw = dict(fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=512)
for i in range(4):
    torch.cuda.synchronize(); time.sleep(0.05)
    t0 = time.perf_counter()
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
    torch.cuda.synchronize()
    print(f"build {i}: {1e3 * (time.perf_counter() - t0):.2f} ms")
