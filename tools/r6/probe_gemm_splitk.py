"""Mid-size GEMMs: the engine's automatic schedule against explicit split-K factors and stream-K."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from curvlinops_amd import _hip
_hip.load()
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return 1e3 * e0.elapsed_time(e1) / n
for (M, N, K) in ((1024, 1024, 1024), (512, 2304, 2304), (512, 2304, 512), (384, 1152, 1152), (256, 2304, 2304), (512, 4608, 512), (2048, 2048, 2048), (2304, 2304, 128)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    res = {"auto": t(lambda: _hip.gemm(A, B, out=out)), "torch": t(lambda: torch.matmul(A, B, out=out))}
    for sk in (1, 2, 3, 4, 6, 8, -1):
        try:
            res[f"sk{sk}"] = t(lambda: _hip.gemm(A, B, out=out, splitk=sk))
        except Exception as e:
            res[f"sk{sk}"] = float("nan")
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{k} {v:6.1f}" for k, v in res.items()), flush=True)
