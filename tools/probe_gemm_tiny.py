"""Scratch: latency of tiny GEMMs (small-network regime, e.g. BASELINE config C1) in a launch chain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
torch.backends.cuda.matmul.allow_tf32 = False
def bench(M, N, K, ta, tb, label):
    A = torch.randn(K, M, device="cuda").T if ta else torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda").T if tb else torch.randn(K, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    res = []
    for fn in (lambda: _hip.gemm(A, B, out=out), lambda: torch.matmul(A, B, out=out)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 300
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    print(f"{label:22s} M={M:5d} N={N:5d} K={K:6d}: clo {res[0]:7.1f} us | torch {res[1]:7.1f} us")
for (M, N, K) in ((64, 256, 128), (64, 64, 256), (64, 256, 64), (64, 128, 256), (128, 256, 64), (256, 128, 64), (64, 10, 64),
                  (16, 256, 256), (32, 512, 512), (64, 1024, 1024), (128, 1024, 1024), (256, 256, 256), (512, 512, 512)):
    bench(M, N, K, False, True, "NT (fwd)")
    bench(M, N, K, False, False, "NN (dprev)")
    bench(M, N, K, True, False, "TN (outer)")
