"""Scratch: split-K sweep on mid-size GEMM shapes (run under different CLO_GEMM_SMALL_MAX)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
lib = _hip.load()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for (M, N, K) in ((128, 2304, 2304), (384, 1152, 1152), (512, 4608, 4608), (512, 2304, 2304), (256, 2304, 2304), (512, 4608, 512), (512, 2304, 512), (1024, 1024, 1024), (2048, 2048, 2048)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    auto = lib.clo_gemm_suggest_splitk(M, N, K, 1)
    row = [f"M={M} N={N} K={K} auto={auto}:"]
    ms = t(lambda: _hip.gemm(A, B, out=out)); row.append(f"AUTO={ms*1e3:.0f}us/{fl/ms/1e9:.0f}TF |")
    for s in (1, 2, 3, 4, 6, 8, 12):
        if K // s < 64: continue
        ms = t(lambda: _hip.gemm(A, B, out=out, splitk=s))
        row.append(f"s{s}={ms*1e3:.0f}us/{fl/ms/1e9:.0f}TF")
    ms = t(lambda: torch.matmul(A, B, out=out)); row.append(f"| torch {ms*1e3:.0f}us/{fl/ms/1e9:.0f}TF")
    print(" ".join(row))
