"""Native eigensolver, one matrix at a time: wall time per order (for the unit-scheduling estimate of eigh_many)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import linalg_native as L
torch.manual_seed(0)
for n in (64, 128, 256, 512, 576, 1152, 2304, 4608):
    X = torch.randn(2 * n, n, device="cuda"); A = X.T @ X / (2 * n)
    L._eigh_full(A); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): L._eigh_full(A)
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 3
    grp = [A.clone() for _ in range(4)]
    L._eigh_native_group(grp); torch.cuda.synchronize()
    t0 = time.perf_counter(); L._eigh_native_group(grp); torch.cuda.synchronize(); t4 = time.perf_counter() - t0
    print(f"n={n}: single {1e3*t1:.2f} ms   group of 4: {1e3*t4:.2f} ms ({1e3*t4/4:.2f} per matrix)")
