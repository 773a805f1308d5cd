"""Per-geometry cost of ResNet-18's factor kernels (B = 512): pixel Gram SYRK, fold, materialised route, small G SYRKs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
B = 512
for (C_, H, W, s) in [(64, 8, 8, 1), (64, 8, 8, 2), (128, 4, 4, 1), (128, 4, 4, 2), (256, 2, 2, 1), (256, 2, 2, 2), (512, 1, 1, 1)]:
    x = torch.randn(B, C_, H, W, device=dev)
    k, st, p, d = (3, 3), (s, s), (1, 1), (1, 1)
    dd = C_ * 9 + 1
    Cm = torch.empty(dd, dd, device=dev)
    n = C_ * H * W
    X2 = x.view(B, n)
    gam = torch.empty(n, n, device=dev)
    OH = (H + 2 - 3) // s + 1
    t_syrk = t(lambda: _hip.syrk_accum(gam, X2, alpha=1.0, beta=0.0))
    cs = X2.sum(0)
    lib = _hip.load()
    stream = torch.cuda.current_stream().cuda_stream
    t_fold = t(lambda: lib.clo_patch_fold_f32(Cm.data_ptr(), dd, gam.data_ptr(), n, cs.data_ptr(), B, C_, H, W, 3, 3, s, s, 1, 1, 1, 1, OH, OH, 1, 1.0, 0.0, stream))
    t_all = t(lambda: _hip.pixel_gram_accum(Cm, x, k, st, p, d, alpha=1.0, beta=0.0, ones_col=True))
    def mat():
        pm = _hip.im2col(x, k, st, p, d)
        _hip.syrk_accum(Cm, pm.reshape(-1, pm.shape[-1]), alpha=1.0, beta=0.0, ones_col=True)
    t_mat = t(mat)
    print(f"conv3x3 C={C_} {H}x{W} s{s}: pixel Gram SYRK [{B}x{n}] {t_syrk:.1f} us, fold -> {dd}^2 {t_fold:.1f} us, whole {t_all:.1f} us | im2col + patch SYRK {t_mat:.1f} us", flush=True)
for rows, d in [(131072, 64), (32768, 64), (8192, 128), (2048, 256), (512, 512), (512, 10), (8192, 65), (2048, 129), (512, 257), (512, 513)]:
    Xg = torch.randn(rows, d if d % 2 == 0 else d - 1, device=dev)
    ones = d % 2 == 1
    Cm = torch.empty(d, d, device=dev)
    print(f"SYRK rows={rows} d={d}: {t(lambda: _hip.syrk_accum(Cm, Xg, alpha=1.0, beta=0.0, ones_col=ones)):.1f} us", flush=True)
