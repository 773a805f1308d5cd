"""Round-5 mid-size sweep: the library's automatic choice against torch.matmul (hipBLASLt) on the shapes the review names --
the round-3 rows, the K-column GEMM of the C2 path, small products of the Cholesky recursion, the SYRK shapes of a
ResNet-18 factor build (clo: symmetric kernel, upper block triangle + mirror; torch: the full X^T X)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
lib = _hip.load()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return 1e3 * e0.elapsed_time(e1) / n
print("# GEMM C[M,N] = A[M,K] B[K,N] (NN): us, TFLOP/s, clo / torch")
for (M, N, K, why) in ((128, 2304, 2304, "r03 row"), (384, 1152, 1152, "r03 row"), (512, 4608, 4608, "r03 row"), (512, 2304, 2304, "r03 row"),
                       (256, 2304, 2304, "r03 row"), (512, 4608, 512, "r03 row"), (512, 2304, 512, "r03 row"), (1024, 1024, 1024, "r03 row"),
                       (2048, 2048, 2048, "r03 row"), (2688, 256, 2688, "K-column chain of C2"), (128, 2688, 2688, "C2 layer, 128 rows"),
                       (128, 128, 1024, "Cholesky panel"), (256, 256, 512, "Cholesky update"), (512, 512, 512, "Cholesky update"),
                       (1024, 128, 128, "Cholesky panel"), (2304, 2304, 128, "rank-128 trailing update"), (4608, 4608, 512, "rank-512 trailing update")):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    us = t(lambda: _hip.gemm(A, B, out=out)); ut = t(lambda: torch.matmul(A, B, out=out))
    print(f"M={M:5d} N={N:5d} K={K:5d} ({why:26s}): clo {us:7.1f} us {fl/us/1e6:6.1f} TF | torch {ut:7.1f} us {fl/ut/1e6:6.1f} TF | clo/torch {us/ut:4.2f}")
print("# SYRK C[d,d] = X^T X, X[rows,d]: clo_syrk_accum_f32 (upper block triangle + mirror) vs torch X.T @ X (full product)")
for (rows, d, why) in ((512, 4096, "pixel Gram, layer1"), (512, 2048, "pixel Gram, layer2"), (512, 1024, "pixel Gram, layer3"), (512, 512, "pixel Gram / G, layer4"),
                       (131072, 64, "G, stem"), (32768, 64, "G, layer1"), (8192, 128, "G, layer2"), (2048, 256, "G, layer3"), (131072, 148, "A, stem (patches)"),
                       (32768, 576, "A, layer1 (patches, round 4)"), (512, 4608, "A, layer4 (patches, round 4)")):
    X = torch.randn(rows, d, device="cuda"); C = torch.empty(d, d, device="cuda")
    fl = 2.0 * rows * d * d
    us = t(lambda: _hip.syrk_accum(C, X, alpha=1.0, beta=0.0)); ut = t(lambda: torch.matmul(X.T, X, out=C))
    print(f"rows={rows:6d} d={d:5d} ({why:28s}): clo {us:7.1f} us | torch {ut:7.1f} us {fl/ut/1e6:6.1f} TF(full) | clo/torch {us/ut:4.2f}")
