"""EKFAC (ResNet-18, 512 rows, joint W+b, 1 MC sample) build time: median of 5 after a warm-up, and the kernel time of
the squared-product kernels in the last build when run under rocprofv3."""
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
ts = []
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    E = C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("ekfac_total ms:", " ".join(f"{t:.1f}" for t in ts), flush=True)
