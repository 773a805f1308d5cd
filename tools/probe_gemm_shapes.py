"""GEMM C[M,N] = A[M,K] B[K,N] (fp32, NN and the W^T-style TN) of whatever library CLO_HIP_LIB names against torch.matmul:
us over 20 back-to-back calls.  Shapes: argv triples M,N,K or the mid-size list of the round-5 sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return 1e3 * e0.elapsed_time(e1) / n
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (128, 2304, 2304), (384, 1152, 1152), (512, 4608, 4608), (512, 2304, 2304), (256, 2304, 2304), (512, 4608, 512), (512, 2304, 512),
    (1024, 1024, 1024), (2048, 2048, 2048), (2688, 256, 2688), (256, 2688, 2688), (128, 2688, 2688), (2304, 2304, 128), (4608, 4608, 512),
    (1536, 1536, 1536), (768, 3072, 768), (3072, 768, 768)]
tag = os.path.basename(os.environ.get("CLO_HIP_LIB", "default"))
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    At = torch.randn(K, M, device="cuda")
    ref = A @ B
    _hip.gemm(A, B, out=out); err = float((out - ref).abs().max() / ref.abs().max())
    us = t(lambda: _hip.gemm(A, B, out=out)); ut = t(lambda: torch.matmul(A, B, out=out))
    ust = t(lambda: _hip.gemm(At.T, B, out=out)); utt = t(lambda: torch.matmul(At.T, B, out=out))
    fl = 2.0 * M * N * K
    print(f"{tag} M={M:5d} N={N:5d} K={K:5d}: NN clo {us:7.1f} us {fl/us/1e6:6.1f} TF torch {ut:7.1f} ratio {us/ut:4.2f} | TN clo {ust:7.1f} torch {utt:7.1f} ratio {ust/utt:4.2f} | err {err:.1e}", flush=True)
