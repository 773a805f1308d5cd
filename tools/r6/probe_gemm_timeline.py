"""Phase timeline of gemm_v3_kernel (-DCLO_V3_TIMING build, tools/buildone.sh v3time gemm_v3.hip -DCLO_V3_TIMING; run with
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_v3time.so): wall_clock64 stamps (100 MHz) of thread 0 of every workgroup:
0 entry, 1 first k tile landed, 2 end of a segment that leaves a partial, 3 partial stored + flag, 4 end of the segment that
finishes a tile, 5 partials of the others added, 6 kernel end."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from curvlinops_amd import _hip
lib = _hip.load()
stamps = torch.zeros(1024, 8, dtype=torch.int64, device="cuda")
lib.clo_v3_timing_set.argtypes = [ctypes.c_void_p]; lib.clo_v3_timing_set.restype = None
lib.clo_v3_timing_set(ctypes.c_void_p(stamps.data_ptr()))
shapes = [(1024, 1024, 1024), (512, 2304, 2304), (256, 2304, 2304), (512, 2304, 512), (384, 1152, 1152), (2048, 2048, 2048), (512, 4608, 4608)]
names = ["entry", "tile0 landed", "seg end (partial)", "partial out+flag", "seg end (finisher)", "partials added", "end", "prologue DMAs issued"]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    for _ in range(5): _hip.gemm(A, B, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _hip.gemm(A, B, out=out)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    stamps.zero_(); torch.cuda.synchronize()
    _hip.gemm(A, B, out=out); torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    live = s[:, 0] > 0
    if not live.any():
        print(f"M={M} N={N} K={K}: {us:.1f} us per call; not served by gemm_v3_kernel"); continue
    t0 = s[live, 0].min()
    print(f"M={M} N={N} K={K}: {us:.1f} us per call; {int(live.sum())} workgroups; stamps in us from the first entry (min / mean / max over the workgroups that took the stamp)")
    for i, nm in enumerate(names):
        m = live & (s[:, i] > 0)
        if not m.any(): continue
        v = (s[m, i] - t0) / 100.0
        print(f"   {i} {nm:20s} n={int(m.sum()):4d}  {v.min():6.2f} {v.mean():6.2f} {v.max():6.2f}")
    both = live & (s[:, 2] > 0) & (s[:, 3] > 0)
    if both.any(): print(f"   partial store + drain: mean {((s[both, 3] - s[both, 2]) / 100.0).mean():.2f} us")
    fin = live & (s[:, 4] > 0) & (s[:, 5] > 0)
    if fin.any():
        print(f"   finisher wait + add:   mean {((s[fin, 5] - s[fin, 4]) / 100.0).mean():.2f} max {((s[fin, 5] - s[fin, 4]) / 100.0).max():.2f} us;  epilogue to end: mean {((s[fin, 6] - s[fin, 5]) / 100.0).mean():.2f} us")
    pro = live & (s[:, 1] > 0)
    print(f"   prologue (entry -> tile 0 landed): mean {((s[pro, 1] - s[pro, 0]) / 100.0).mean():.2f} max {((s[pro, 1] - s[pro, 0]) / 100.0).max():.2f} us")
