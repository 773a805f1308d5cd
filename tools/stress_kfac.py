import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
import torch
from torch import nn
import curvlinops_amd as C
from models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18().to(dev).eval(); params = kfac_params(model); B = 64
X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="type-2", separate_weight_and_bias=False,
                         check_deterministic=False, num_data=B)
v = torch.rand(K.shape[1], device=dev)
ref = None
t0 = time.perf_counter()
for i in range(40):
    Kinv = K.inverse(damping=1e-2)
    out = Kinv @ v
    if ref is None: ref = out.clone()
    d = float((out - ref).abs().max() / ref.abs().max())
    assert d < 1e-5 and torch.isfinite(out).all(), (i, d)
torch.cuda.synchronize()
print(f"40 inverse builds + products identical to 1e-5: ok, {(time.perf_counter()-t0)/40*1e3:.1f} ms each")
E = C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="type-2", separate_weight_and_bias=False,
                          check_deterministic=False, num_data=B)
r0 = E @ v
for i in range(5):
    E2 = C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="type-2", separate_weight_and_bias=False,
                               check_deterministic=False, num_data=B)
    d = float((E2 @ v - r0).abs().max() / r0.abs().max())
    assert d < 1e-3, (i, d)
print("5 EKFAC rebuilds consistent: ok")
