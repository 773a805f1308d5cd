"""KFAC / EKFAC matvec of ResNet-18 (C4 factors): all blocks in ONE foreign call (clo_kron_matmat_blocks) against the
rounds 1-3 composition in Python (batched groups of equal shapes + a stream pool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd.kronecker import BlockDiagonalLinearOperator
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18(num_classes=10).to(dev).eval(); params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for name, cls in (("KFAC", C.KFACLinearOperator), ("EKFAC", C.EKFACLinearOperator)):
    op = cls(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", mc_samples=1, separate_weight_and_bias=False, check_deterministic=False)
    for K in (1, 8, 32):
        V = torch.rand(op.shape[1], K, device=dev) if K > 1 else torch.rand(op.shape[1], device=dev)
        res = {}
        for flag in (True, False):
            BlockDiagonalLinearOperator.SINGLE_CALL = flag
            BlockDiagonalLinearOperator.GROUP_FIRST = not flag
            res[flag] = (t(lambda: op @ V), op @ V)
        err = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
        print(f"{name} @ [D, {K}]: single call {res[True][0]:.3f} ms | python composition {res[False][0]:.3f} ms | rel diff {err:.1e}", flush=True)
