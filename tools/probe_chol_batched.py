"""Scratch: batched damped Cholesky inverse (equal-size factors in one chain) per group."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
dev = torch.device("cuda:0")
for n, b in ((64, 8), (512, 5), (577, 5), (1153, 4), (2305, 4), (4609, 3), (768, 60), (3072, 24)):
    mats = []
    for _ in range(min(b, 4)):
        X = torch.randn(n + 8, n, device=dev); mats.append(X.T @ X / n)
    mats = [mats[i % len(mats)] for i in range(b)]
    outs = [torch.empty(n, n, device=dev) for _ in range(b)]
    status = torch.zeros(b, device=dev, dtype=torch.int32)
    def run(): _hip.cholesky_inverse_batched_into(mats, [1e-3] * b, outs, status)
    def one(): _hip.cholesky_inverse_async(mats[0], 1e-3)
    res = []
    for fn in (run, one):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 3 * 1e3)
    print(f"n={n:5d} batch={b:2d}: batched chain {res[0]:8.2f} ms ({res[0]/b:6.2f} per factor) | single factor {res[1]:7.2f} ms")
