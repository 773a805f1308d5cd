"""Scratch: streaming tall-skinny Gram kernel vs split-K symmetric GEMM, by width."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
lib = _hip.load()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for d in (8, 16, 32, 48, 64, 96, 128):
    for rows in (8192, 32768, 131072, 524288):
        X = torch.randn(rows, d, device="cuda"); C = torch.empty(d, d, device="cuda")
        if not lib.clo_gram_tall_supported(rows, d, 0): continue
        tg = t(lambda: _hip.syrk_accum(C, X, alpha=1.0, beta=0.0))
        best = min((t(lambda: _hip.syrk_accum(C, X, alpha=1.0, beta=0.0, splitk=s)), s) for s in (16, 32, 64) if rows // s >= 64)
        print(f"d {d:4d} rows {rows:7d}: gram_tall {tg:7.1f} us | gemm best {best[0]:7.1f} us (s={best[1]})")
