"""ResNet-18 KFAC factors of two half shards vs the full batch (float32 GPU): actual relative differences per
factor, and against a float64 GPU run of the same."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, copy
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
B = 512
X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
kw = dict(fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False)
lf = nn.CrossEntropyLoss()
def factors(m, X, y, **extra):
    K = C.KFACLinearOperator(m, lf, kfac_params(m), [(X, y)], **kw, **extra)
    return [S for blk in K[1] for S in blk]
full = factors(model, X, y)
h0 = factors(model, X[:B // 2], y[:B // 2], num_data=B); h1 = factors(model, X[B // 2:], y[B // 2:], num_data=B)
m64 = copy.deepcopy(model).double()
full64 = factors(m64, X.double(), y)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))
print("idx type n | shards-vs-full(fp32)  full fp32-vs-fp64  shards(fp32)-vs-fp64")
for i, (S, a, b, S64) in enumerate(zip(full, h0, h1, full64)):
    print(f"{i:3d} {'G' if i % 2 == 0 else 'A'} {S.shape[0]:5d} | {rel(a + b, S):.1e}   {rel(S, S64):.1e}   {rel(a + b, S64):.1e}")
