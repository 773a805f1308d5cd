"""Scratch: throughput of clo_gemm_f32 on the shapes the large-batch MLP path and KFAC use."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
def bench(M, N, K, ta, tb, label):
    A = torch.randn(K, M, device="cuda").T if ta else torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda").T if tb else torch.randn(K, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(3): _hip.gemm(A, B, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): _hip.gemm(A, B, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    e0.record()
    for _ in range(n): torch.matmul(A, B, out=out)
    e1.record(); torch.cuda.synchronize()
    ms_t = e0.elapsed_time(e1) / n
    fl = 2.0 * M * N * K
    print(f"{label:28s} M={M:5d} N={N:5d} K={K:6d}: clo {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF | torch {ms_t*1e3:8.1f} us {fl/ms_t/1e9:6.1f} TF")
torch.backends.cuda.matmul.allow_tf32 = False
bench(4096, 4096, 4096, False, False, "square NN")
bench(4096, 4096, 4096, False, True, "square NT")
bench(8192, 8192, 8192, False, True, "square NT")
for n in (128, 512, 2048):
    bench(n, 2688, 2688, False, True, f"fwd  N={n} (A W^T)")
    bench(n, 2688, 2688, False, False, f"bwd  N={n} (D W)")
    bench(2688, 2688, n, True, False, f"outer N={n} (D^T A)")
bench(4608, 4608, 512, True, False, "syrk-like 4608 rows=512")
bench(576, 576, 32768, True, False, "syrk-like 576 rows=32768")
