"""Kernel-only durations of the mid-size GEMM sweep (run under rocprofv3 --kernel-trace; summarised by
tools/r6/gemm_kernel_time_summary.py): per shape a marker fill, 10 calls of the library's GEMM, a marker, 10 calls of torch.matmul
(hipBLASLt), so that the Python / dispatch cost per call does not enter the comparison."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from curvlinops_amd import _hip
_hip.load()
SHAPES = ((128, 2304, 2304), (384, 1152, 1152), (512, 4608, 4608), (512, 2304, 2304), (256, 2304, 2304), (512, 4608, 512), (512, 2304, 512),
          (1024, 1024, 1024), (2048, 2048, 2048), (2688, 256, 2688), (128, 2688, 2688), (2304, 2304, 128), (4608, 4608, 512))
mark = torch.empty(54321, device="cuda")
import time
for (M, N, K) in SHAPES:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    for _ in range(3): _hip.gemm(A, B, out=out); torch.matmul(A, B, out=out)
    torch.cuda.synchronize()
    mark.fill_(1.0)
    t0 = time.perf_counter()
    for _ in range(10): _hip.gemm(A, B, out=out)
    t1 = time.perf_counter()
    mark.fill_(2.0)
    for _ in range(10): torch.matmul(A, B, out=out)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"SHAPE {M} {N} {K} host_us_per_call clo {1e5 * (t1 - t0):.1f} torch {1e5 * (t2 - t1):.1f}")
mark.fill_(3.0); torch.cuda.synchronize()
