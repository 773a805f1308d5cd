import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for d in (4608, 4609, 2304, 2305, 576, 577):
    G = torch.randn(512, 512, device="cuda"); X = torch.randn(512, d, device="cuda"); A = torch.randn(d, d, device="cuda")
    t1 = t(lambda: _hip.gemm(G, X)); t2 = t(lambda: _hip.gemm(X, A.T))
    print(f"d_in'={d}: G X {t1:.0f} us, (G X) A^T {t2:.0f} us")
