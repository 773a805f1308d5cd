"""Phase timeline of potrf_node128_kernel (-DCLO_POTRF_TIMING build): thread 0 of block 0 stamps wall_clock64 (100 MHz) at entry, after the
load, and per 16-column step after the diagonal block, after the panel and after the trailing update; then after the off-diagonal blocks of
L^-1 and after the stores.  n = 128: the Cholesky inverse of a 128 x 128 matrix is one such node."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from curvlinops_amd import _hip
lib = _hip.load()
stamps = torch.zeros(64, dtype=torch.int64, device="cuda")
lib.clo_potrf_timing_set.argtypes = [ctypes.c_void_p]; lib.clo_potrf_timing_set.restype = None
lib.clo_potrf_timing_set(ctypes.c_void_p(stamps.data_ptr()))
torch.manual_seed(0)
X = torch.randn(512, 128, device="cuda")
A = X.T @ X / 512
for _ in range(5): _hip.cholesky_inverse(A, 1e-3)
torch.cuda.synchronize()
rows = []
for rep in range(5):
    stamps.zero_(); torch.cuda.synchronize()
    _hip.cholesky_inverse(A, 1e-3); torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.int64)
    rows.append((s - s[0]) / 100.0)
r = np.median(np.array(rows), axis=0)
print(f"load done {r[1]:.2f} us")
prev = r[1]
tot = {"diag": 0.0, "panel": 0.0, "trail": 0.0}
for kb in range(8):
    d, p, t = r[2 + 3 * kb], r[3 + 3 * kb], r[4 + 3 * kb]
    print(f"  step {kb}: diagonal block {d - prev:5.2f}  panel {p - d:5.2f}  trailing update {t - p:5.2f}   (at {t:6.2f} us)")
    tot["diag"] += d - prev; tot["panel"] += p - d; tot["trail"] += t - p
    prev = t
print(f"sum: diagonal blocks {tot['diag']:.2f}, panels {tot['panel']:.2f}, trailing updates {tot['trail']:.2f} us")
print(f"off-diagonal blocks of L^-1 {r[26] - prev:.2f} us; stores {r[27] - r[26]:.2f} us; kernel {r[27]:.2f} us (stamps included)")
