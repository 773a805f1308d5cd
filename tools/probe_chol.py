"""Scratch: damped Cholesky inverse per factor size (alone on one stream) vs torch (rocSOLVER)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
dev = torch.device("cuda:0")
for n in (64, 128, 256, 512, 577, 1152, 1153, 2304, 2305, 4608, 4609):
    X = torch.randn(2 * n, n, device=dev)
    A = X.T @ X / (2 * n)
    def ours(): return _hip.cholesky_inverse_async(A, 1e-3)[0]
    def ref():
        L = torch.linalg.cholesky(A + 1e-3 * torch.eye(n, device=dev))
        return torch.cholesky_inverse(L)
    res = []
    for fn in (ours, ref):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 5
        for _ in range(reps): o = fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / reps * 1e3)
    err = float((ours() - ref()).abs().max() / ref().abs().max())
    print(f"n={n:5d}: clo {res[0]:8.3f} ms ({n**3/res[0]/1e9:6.2f} TF) | torch {res[1]:8.3f} ms | rel diff {err:.1e}")
