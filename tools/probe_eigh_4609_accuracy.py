"""Accuracy of the native eigensolver on the rank-deficient 4609 test matrix (reconstruction / orthogonality)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import linalg_native as L
for n in (2305, 4609):
    g = torch.Generator().manual_seed(n)
    X = torch.rand(max(16, n // 3), n, generator=g, dtype=torch.float64)
    A64 = X.T @ X / X.shape[0]
    A = A64.cuda().float()
    lam, Q = L.eigh(A)
    Qd, ld_ = Q.double().cpu(), lam.double().cpu()
    orth = float((Qd.T @ Qd - torch.eye(n, dtype=torch.float64)).abs().max())
    rec = float(((Qd * ld_) @ Qd.T - A.double().cpu()).abs().max()) / float(A64.abs().max())
    print(f"n={n}: orth {orth:.2e}  rec/|A|max {rec:.2e}  lib={os.environ.get('CLO_HIP_LIB','default')[-20:]}")
