"""ResNet-18 KFAC: wall time of successive `K.inverse(damping)` calls (all 42 damped Cholesky inverses)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18().to(dev).eval(); params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False,
                         check_deterministic=False)
ts = []
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Kinv = K.inverse(damping=1e-3)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"CLO_CHOL_PIPE={os.environ.get('CLO_CHOL_PIPE', '1')} CLO_INV_STREAMS={os.environ.get('CLO_INV_STREAMS', '4')}: "
      + " ".join(f"{t:.1f}" for t in ts) + " ms", flush=True)
