"""Cost of the orthogonality verification inside eigh_many on ResNet-18's 42 KFAC factors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import linalg_native as L
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False,
                         check_deterministic=False, num_data=512)
facs = [S for blk in K[1] for S in blk._factors]
def t(fn, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print(f"eigh_many with verification   : {t(lambda: L.eigh_many(facs)):.1f} ms")
L._ORTH_TOL = float("inf"); orig = L._orth_defect
L._orth_defect = lambda Q: Q.new_zeros(Q.shape[:-2])
print(f"eigh_many without verification: {t(lambda: L.eigh_many(facs)):.1f} ms")
L._orth_defect = orig; L._ORTH_TOL = 1e-4
t0 = time.perf_counter(); E = C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=512); torch.cuda.synchronize()
print(f"EKFAC build (1st) {1e3*(time.perf_counter()-t0):.1f} ms")
print(f"EKFAC build (min of 3) {t(lambda: C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type='mc', separate_weight_and_bias=False, check_deterministic=False, num_data=512), 3):.1f} ms")
print("orthogonality defect of plain torch.linalg.eigh per factor (normalised input):")
for S in facs:
    An, s = L._unit_scale(S)
    Q = torch.linalg.eigh(An).eigenvectors
    d = float(L._orth_defect(Q))
    if d > 2e-5:
        print(f"   n={S.shape[0]}: |Q^T Q - I| = {d:.2e}   zero rows: {int((S.abs().sum(1) == 0).sum())}")
print("done")
