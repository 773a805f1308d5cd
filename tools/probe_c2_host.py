"""Host cost of one C2 GGN product through the operator API: enqueue time of 300 products (no synchronisation) against the
time until the GPU has finished them.  The step time of bench.py is max(host, kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C

dev = torch.device("cuda:0")
torch.manual_seed(0)
dims = [1024, 2688, 2688, 10]
model = nn.Sequential(nn.Linear(dims[0], dims[1]), nn.ReLU(), nn.Linear(dims[1], dims[2]), nn.ReLU(), nn.Linear(dims[2], dims[3])).to(dev)
params = dict(model.named_parameters())
X, y = torch.rand(8, dims[0], device=dev), torch.rand(8, dims[3], device=dev)
op = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
vs = [torch.rand(op.shape[1], device=dev) for _ in range(8)]
for _ in range(20):
    op @ vs[0]
torch.cuda.synchronize()
for n in (20, 300):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            op @ vs[i & 7]
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"n={n}: host enqueue {1e6*(t1-t0)/n:.1f} us/product, until done {1e6*(t2-t0)/n:.1f} us/product", flush=True)
# the pieces of the host path
import timeit
nat = op._native
print("is_current", 1e6 * min(timeit.repeat(lambda: nat.is_current(op._params), number=1000, repeat=5)) / 1000, "us")
print("flat_key", 1e6 * min(timeit.repeat(lambda: op._native_flat_key(), number=1000, repeat=5)) / 1000, "us")
print("flat_setup", 1e6 * min(timeit.repeat(lambda: op._native_flat_setup(), number=1000, repeat=5)) / 1000, "us")
print("empty_like", 1e6 * min(timeit.repeat(lambda: torch.empty_like(vs[0]), number=1000, repeat=5)) / 1000, "us")
