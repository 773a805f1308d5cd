"""Scratch: GEMM shapes of the ResNet-18 Kronecker matvec (G V A with V = [d_out, d_in])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
torch.backends.cuda.matmul.allow_tf32 = False
def bench(M, N, K, label):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    res = []
    for fn in (lambda: _hip.gemm(A, B, out=out), lambda: torch.matmul(A, B, out=out)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30; e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1) / n)
    fl = 2.0 * M * N * K
    print(f"{label:26s} M={M:5d} N={N:5d} K={K:5d}: clo {res[0]*1e3:8.1f} us {fl/res[0]/1e9:6.1f} TF | torch {res[1]*1e3:8.1f} us {fl/res[1]/1e9:6.1f} TF")
for do, di in ((64, 576), (128, 576), (128, 1152), (256, 1152), (256, 2304), (512, 2304), (512, 4608), (1000, 512)):
    bench(do, di, di, f"V A   ({do}x{di})")
    bench(do, di, do, f"G V   ({do}x{di})")
