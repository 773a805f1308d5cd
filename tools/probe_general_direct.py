"""General nets (torch.func path): single-column products with and without the vmap over a length-one axis."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import curvature
from benchmarks.models import Encoder, ResNet18
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2]
cases = []
enc = Encoder().to(dev).eval(); Xe = torch.rand(8, 128, 768, device=dev); ye = torch.randint(0, 10, (8,), device=dev)
cases.append(("encoder 85M", enc, Xe, ye))
rn = ResNet18(num_classes=10).to(dev).eval(); Xr = torch.rand(128, 3, 32, 32, device=dev); yr = torch.randint(0, 10, (128,), device=dev)
cases.append(("resnet18 b128", rn, Xr, yr))
for name, model, X, y in cases:
    params = dict(model.named_parameters())
    for cls in (C.EFLinearOperator, C.GGNLinearOperator, C.HessianLinearOperator):
        op = cls(model, nn.CrossEntropyLoss(), params, [(X, y)], check_deterministic=False)
        v = torch.rand(op.shape[1], device=dev)
        res = {}
        for mode in ("one pass", "direct", "vmap"):
            curvature.CurvatureLinearOperator.SINGLE_COLUMN_DIRECT = mode != "vmap"
            curvature.FUSED_SINGLE_COLUMN = mode == "one pass"
            res[mode] = (t(lambda: op @ v), op @ v)
        err = max(float((res[m][1] - res["vmap"][1]).abs().max() / res["vmap"][1].abs().max()) for m in res)
        print(f"{name} {cls.__name__}: one forward pass {res['one pass'][0]:.1f} ms | direct {res['direct'][0]:.1f} ms | vmap {res['vmap'][0]:.1f} ms | max rel diff {err:.1e}", flush=True)
