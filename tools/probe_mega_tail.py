"""Tail of the C2 product time of whatever library CLO_HIP_LIB names: BATCHES batches of 20 products, each batch timed on
the host around a synchronize; prints median / p99 / max batch time per product and the number of batches more than 5 us
per product above the median (a poll that stalled shows up here, not in a 300-product mean)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
B, n = 20, int(os.environ.get("BATCHES", "3000"))
def run(m):
    for i in range(m):
        k = i % nv
        plan.ggn_matvec(W, b, VW[k], Vb[k], OW[k], Ob[k], X, 0, 2.0 / 80, 1.0, 0.0)
run(100); torch.cuda.synchronize()
ts = np.empty(n)
for j in range(n):
    t0 = time.perf_counter(); run(B); torch.cuda.synchronize(); ts[j] = 1e6 * (time.perf_counter() - t0) / B
med = np.median(ts)
print(f"{os.path.basename(os.environ.get('CLO_HIP_LIB', 'default'))}: batches of {B}: median {med:.2f} p99 {np.percentile(ts, 99):.2f} "
      f"max {ts.max():.2f} us per product; {int((ts > med + 5).sum())} of {n} batches > median + 5 us; "
      f"persistent status {_hip.persistent_status() if hasattr(_hip, 'persistent_status') else '?'}", flush=True)
