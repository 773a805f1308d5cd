"""Scratch: one damped Cholesky inverse size, for rocprof kernel breakdowns."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
X = torch.randn(2 * n, n, device=dev)
A = X.T @ X / (2 * n)
for _ in range(2): _hip.cholesky_inverse_async(A, 1e-3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): _hip.cholesky_inverse_async(A, 1e-3)
torch.cuda.synchronize(); print(f"n={n}: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
