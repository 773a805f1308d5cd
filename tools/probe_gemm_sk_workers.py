import sys, os
sys.path.insert(0, "/root/repo")
import torch
from curvlinops_amd import _hip
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
M, N, K = [int(x) for x in sys.argv[1:4]]
A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
ms = t(lambda: _hip.gemm(A, B, out=out, splitk=-1))
print(f"M={M} N={N} K={K} workers={os.environ.get('CLO_V3_SK_WORKERS')}: {ms*1e3:.1f} us")
