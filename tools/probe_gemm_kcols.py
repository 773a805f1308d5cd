"""GEMM shapes of the K-column chain (C2, 8 rows x K columns): [2688 x 2688] [2688 x 8K], split-K sweep through clo_gemm_f32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
dev = torch.device("cuda:0")
W = torch.randn(2688, 2688, device=dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
for NK in (64, 256, 512):
    B = torch.randn(2688, NK, device=dev); out = torch.empty(2688, NK, device=dev)
    ref = W @ B
    for tr, A in (("W", W), ("W^T", W.T)):
        row = []
        for s in (None, 1, 2, 3, 4, 6, 8, 12, -1):
            try:
                us = t(lambda: _hip.gemm(A, B, out=out, splitk=s))
                err = float((out - (A @ B)).abs().max() / ref.abs().max())
                row.append(f"{'auto' if s is None else s}:{us:6.1f}" + ("" if err < 1e-4 else f"(err {err:.1e})"))
            except Exception as e:
                row.append(f"{s}:fail")
        tt = t(lambda: torch.matmul(A, B, out=out))
        print(f"{tr:4s} x [2688 x {NK:3d}]  " + "  ".join(row) + f"   | torch (hipBLASLt) {tt:6.1f} us")
