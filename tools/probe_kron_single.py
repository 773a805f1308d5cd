"""clo_kron_matmat on one block with an odd-order second factor (joint weight + bias): us per product."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
dev = torch.device("cuda:0")
for (A, B, K) in ((512, 4609, 1), (512, 4609, 8), (256, 2305, 1), (128, 1153, 8), (512, 4608, 1)):
    S1 = torch.rand(A, A, device=dev); S2 = torch.rand(B, B, device=dev); X = torch.rand(K, A * B, device=dev)
    for _ in range(3): _hip.kron_matmat(S1, S2, X, K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): _hip.kron_matmat(S1, S2, X, K)
    torch.cuda.synchronize()
    print(f"{os.path.basename(os.environ.get('CLO_HIP_LIB', 'default'))}: A={A} B={B} K={K}: {1e6 * (time.perf_counter() - t0) / 10:.1f} us", flush=True)
