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
E = C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
for op, name in ((E, "ekfac"), (K, "kfac")):
    for k in (1, 8):
        V = torch.rand(op.shape[1], device=dev) if k == 1 else torch.rand(op.shape[1], k, device=dev)
        for _ in range(3): op @ V
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): op @ V
        torch.cuda.synchronize(); print(f"{name} K={k}: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms", flush=True)
