"""ResNet-18 (full size, 128 rows): KFAC / EKFAC products and damped inverses in float32 (native path) against the same
package in float64 on the GPU (torch path)."""
import os, sys, copy, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m32 = ResNet18().to(dev).eval(); m64 = copy.deepcopy(m32).double()
B = 128
X = torch.rand(B, 3, 32, 32, device=dev); y = torch.randint(0, 10, (B,), device=dev)
kw = dict(fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False)
lf = nn.CrossEntropyLoss()
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
for cls in (C.KFACLinearOperator, C.EKFACLinearOperator):
    K32 = cls(m32, lf, kfac_params(m32), [(X, y)], **kw)
    K64 = cls(m64, lf, kfac_params(m64), [(X.double(), y)], **kw)
    v = torch.rand(K32.shape[1], 2, device=dev) - 0.5
    print(f"{cls.__name__}: @ V rel err {rel(K32 @ v, K64 @ v.double()):.1e}", flush=True)
    if cls is C.KFACLinearOperator:
        for mode in ({}, {"use_heuristic_damping": True}, {"use_exact_damping": True}):
            d = 1e-2
            print(f"   inverse(damping={d}, {mode}) @ V rel err {rel(K32.inverse(damping=d, **mode) @ v, K64.inverse(damping=d, **mode) @ v.double()):.1e}", flush=True)
    else:
        print(f"   inverse(damping=1e-2) @ V rel err {rel(K32.inverse(damping=1e-2) @ v, K64.inverse(damping=1e-2) @ v.double()):.1e}", flush=True)
