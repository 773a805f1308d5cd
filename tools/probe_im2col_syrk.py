"""Scratch: fused im2col->SYRK vs materialised patches + SYRK for the conv geometries of ResNet-18 / LeNet-5."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
dev = "cuda"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
geoms = [("stem 3x32x32 k7 s2 p3", 3, 32, 7, 2, 3), ("l1 64x8x8 k3", 64, 8, 3, 1, 1), ("l2.0 64x8x8 k3 s2", 64, 8, 3, 2, 1),
         ("l2 128x4x4 k3", 128, 4, 3, 1, 1), ("l3.0 128x4x4 k3 s2", 128, 4, 3, 2, 1), ("l3 256x2x2 k3", 256, 2, 3, 1, 1),
         ("l4.0 256x2x2 k3 s2", 256, 2, 3, 2, 1), ("l4 512x1x1 k3", 512, 1, 3, 1, 1), ("ds 64x8x8 k1 s2", 64, 8, 1, 2, 0),
         ("lenet1 1x32x32 k5", 1, 32, 5, 1, 0), ("lenet2 6x14x14 k5", 6, 14, 5, 1, 0)]
for name, C, H, k, s, p in geoms:
    x = torch.randn(B, C, H, H, device=dev)
    d = C * k * k
    Cm = torch.empty(d, d, device=dev)
    def mat():
        P = _hip.im2col(x, (k, k), (s, s), (p, p), (1, 1))
        _hip.syrk_accum(Cm, P.reshape(-1, d), alpha=1.0, beta=0.0)
    def fus():
        _hip.im2col_syrk_accum(Cm, x, (k, k), (s, s), (p, p), (1, 1), alpha=1.0, beta=0.0)
    OH = (H + 2 * p - k) // s + 1
    rows = B * OH * OH
    tm, tf = t(mat), t(fus)
    print(f"{name:24s} rows {rows:7d} d {d:5d} patches {rows*d*4/1e6:7.1f} MB: materialised {tm:7.1f} us  fused {tf:7.1f} us  "
          f"({2.0*rows*d*d/tf/1e6:6.1f} TF/s fused, {2.0*rows*d*d/tm/1e6:6.1f} mat)")
