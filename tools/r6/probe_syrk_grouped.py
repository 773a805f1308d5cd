"""The gradient covariances of a ResNet-18 batch (C4, 512 rows): clo_syrk_grouped_f32 (one launch) against one
clo_syrk_accum_f32 per factor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from curvlinops_amd import _hip
_hip.load()
shapes = [(131072, 64)] + [(32768, 64)] * 4 + [(8192, 128)] * 5 + [(2048, 256)] * 5 + [(512, 512)] * 5
Xs = [torch.randn(r, d, device="cuda") for r, d in shapes]
Cs = [torch.empty(d, d, device="cuda") for _, d in shapes]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return 1e3 * e0.elapsed_time(e1) / n
fl = sum(2.0 * r * d * d for r, d in shapes)
ug = t(lambda: _hip.syrk_grouped(Cs, Xs, [1.0] * len(Xs), [0.0] * len(Xs)))
us = t(lambda: [_hip.syrk_accum(c, x, alpha=1.0, beta=0.0) for c, x in zip(Cs, Xs)])
print(f"{os.environ.get('CLO_HIP_LIB', 'default')}: grouped {ug:7.1f} us ({fl / ug / 1e6 / 2:5.1f} TF executed-half) | separate {us:7.1f} us")
