"""Scratch: tall mid-width SYRKs of the ResNet-18 KFAC build (rows x d), split-K sweep vs torch."""
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
for rows, d in [(32768, 576), (8192, 576), (8192, 1152), (2048, 1152), (2048, 2304), (512, 2304), (512, 4608), (131072, 64), (32768, 64)]:
    X = torch.randn(rows, d, device="cuda")
    C = torch.empty(d, d, device="cuda")
    auto = lib.clo_syrk_suggest_splitk(d, rows)
    row = [f"rows {rows:6d} d {d:5d} auto={auto:2d}:"]
    tall = lib.clo_gram_tall_supported(rows, d, 0)
    row.append(f"default {t(lambda: _hip.syrk_accum(C, X, alpha=1.0, beta=0.0)):7.1f} us{' (gram_tall)' if tall else ''} |")
    for s in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64):
        if rows // s < 64: continue
        row.append(f"s{s}={t(lambda: _hip.syrk_accum(C, X, alpha=1.0, beta=0.0, splitk=s)):.0f}")
    row.append(f"| torch X^T X {t(lambda: torch.matmul(X.T, X, out=C)):.0f} us")
    print(" ".join(row), f" full-flop TF at default: {2.0*rows*d*d/1e6/t(lambda: _hip.syrk_accum(C, X, alpha=1.0, beta=0.0)):.0f}")
