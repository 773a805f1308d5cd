"""Scratch: GEMMs with a huge M and small N = K (rotations of tall blocks)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
for M, N, K in [(85_000_000, 32, 32), (524_288, 64, 64), (8_028_160, 6, 6), (524_288, 576, 576), (1_048_576, 128, 128), (4_000_000, 16, 16)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(2): _hip.gemm(A, B, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): _hip.gemm(A, B, out=out)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    for _ in range(2): torch.matmul(A, B, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): torch.matmul(A, B, out=out)
    torch.cuda.synchronize(); tt = (time.perf_counter() - t0) / n
    byts = 4.0 * M * (K + N)
    print(f"gemm M={M:9d} N={N:4d} K={K:4d}: clo {t*1e3:8.3f} ms ({byts/t/1e12:.2f} TB/s, {2*M*N*K/t/1e12:.1f} TF) | torch {tt*1e3:8.3f} ms")
    del A, out
