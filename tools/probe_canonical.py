"""Fused canonical pack / unpack (clo_canonical_pack_f32) against cat + transpose / transpose + slices, and a
K-column KFAC product of ResNet-18 (bias-carrying convolutions: every group is a joint (W, b) group)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import _hip, canonical
_hip.load()
dev = torch.device("cuda:0")

def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

for rows, cols, K in ((512, 4608, 8), (512, 4608, 32), (64, 576, 32), (256, 2304, 16)):
    w = torch.rand(rows * cols, K, device=dev); b = torch.rand(rows, K, device=dev)
    def old_in():
        j = torch.cat([w.view(rows, cols, K), b.unsqueeze(1)], dim=1).flatten(end_dim=-2)
        return _hip.transpose(j)
    def new_in(): return _hip.canonical_pack(w, b, rows, cols)
    y = new_in()
    def old_out():
        j = _hip.transpose(y).reshape(rows, cols + 1, K)
        return j[:, :cols].reshape(rows, cols, K), j[:, cols].reshape(rows, K)
    def new_out(): return _hip.canonical_unpack(y, rows, cols, True)
    mb = 2 * 4 * rows * (cols + 1) * K / 1e6
    a, bb, c, d = t_us(old_in), t_us(new_in), t_us(old_out), t_us(new_out)
    print(f"rows {rows} cols {cols} K {K}: pack {a:7.1f} -> {bb:7.1f} us ({mb / bb * 1e6 / 1e6:.2f} TB/s) | unpack {c:7.1f} -> {d:7.1f} us ({mb / d:.2f} TB/s)", flush=True)

from benchmarks.models import ResNet18, kfac_params
torch.manual_seed(0)
model = ResNet18().to(dev).eval(); params = kfac_params(model)
X, y = torch.rand(128, 3, 32, 32, device=dev), torch.randint(0, 10, (128,), device=dev)
Kop = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="empirical",
                           separate_weight_and_bias=False, check_deterministic=False)
for K in (1, 8, 32):
    V = torch.rand(Kop.shape[1], K, device=dev)
    print(f"KFAC @ V, K = {K}: {t_us(lambda: Kop @ V, 10) / 1e3:.3f} ms", flush=True)
    if K > 1:
        canonical.KMAJOR_BLOCKS = False
        print(f"   (cat / transpose route: {t_us(lambda: Kop @ V, 10) / 1e3:.3f} ms)", flush=True)
        canonical.KMAJOR_BLOCKS = True
