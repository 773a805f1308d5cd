"""12-layer d = 768 encoder (full size), 4 sequences of 32 tokens: KFAC / EKFAC of the Linear layers in float32 (native)
against float64 (torch path) on the GPU; weight sharing over the sequence (expand)."""
import os, sys, copy, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import Encoder, kfac_params
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m32 = Encoder().to(dev).eval(); m64 = copy.deepcopy(m32).double()
X = torch.rand(4, 32, 768, device=dev); y = torch.randint(0, 10, (4,), device=dev)
kw = dict(fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False)
lf = nn.CrossEntropyLoss()
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
t0 = time.time()
for cls in (C.KFACLinearOperator, C.EKFACLinearOperator):
    K32 = cls(m32, lf, kfac_params(m32), [(X, y)], **kw)
    K64 = cls(m64, lf, kfac_params(m64), [(X.double(), y)], **kw)
    v = torch.rand(K32.shape[1], 2, device=dev) - 0.5
    print(f"{cls.__name__}: D = {K32.shape[1]}  @ V rel err {rel(K32 @ v, K64 @ v.double()):.1e}", flush=True)
    i32 = K32.inverse(damping=1e-2) @ v; i64 = K64.inverse(damping=1e-2) @ v.double()
    print(f"   inverse(1e-2) @ V rel err {rel(i32, i64):.1e}   ({time.time()-t0:.0f} s)", flush=True)
