"""us per C2 GGN product (8 rows, persistent kernel) of whatever library CLO_HIP_LIB names: 3 x 300 products on rotating
40 MB vectors (bench.py's default protocol), straight through the plan (no operator layer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip

dims, acts = [1024, 2688, 2688, 10], [1, 1, 0]
torch.manual_seed(0)
W = [torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5 for i in range(3)]
b = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(3)]
nv = 8
VW = [[torch.rand_like(w) for w in W] for _ in range(nv)]
Vb = [[torch.rand_like(x) for x in b] for _ in range(nv)]
OW = [[torch.empty_like(w) for w in W] for _ in range(nv)]
Ob = [[torch.empty_like(x) for x in b] for _ in range(nv)]
plan = _hip.MLPPlan(dims, acts)
X = torch.rand(8, dims[0], device="cuda")
def run(n):
    for i in range(n):
        k = i % nv
        plan.ggn_matvec(W, b, VW[k], Vb[k], OW[k], Ob[k], X, 0, 2.0 / 80, 1.0, 0.0)
run(30); torch.cuda.synchronize()
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(300); torch.cuda.synchronize(); ts.append(1e6 * (time.perf_counter() - t0) / 300)
print(f"{os.path.basename(os.environ.get('CLO_HIP_LIB', 'default'))}: " + " ".join(f"{t:.2f}" for t in ts) + " us per product", flush=True)
