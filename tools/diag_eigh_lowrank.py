"""Extremely rank-deficient PSD matrices (r rows, n features): torch.linalg.eigh and eigh_sytrd, float32 on the GPU:
residual, orthogonality, and the quantity an exact-damping inverse needs: |Q f(lam) Q^T x - (A + d)^-1 x|."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from curvlinops_amd import linalg_native as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
for n, r in ((845, 10), (845, 2), (845, 100), (300, 3), (1200, 8), (64, 2), (845, 845)):
    X = torch.randn(r, n, dtype=torch.float64)
    A64 = X.T @ X / r
    A = A64.float().to(dev)
    A32 = A.double().cpu()
    d = 1e-2 * float(A32.abs().max())
    x = torch.randn(n, 3, dtype=torch.float64)
    ref = torch.linalg.solve(A32 + d * torch.eye(n, dtype=torch.float64), x)
    for name, fn in (("torch(normalised)", lambda M: L._torch_eigh_scaled(M)), ("torch(raw)", lambda M: tuple(torch.linalg.eigh(M))),
                     ("sytrd", L.eigh_sytrd)):
        lam, Q = fn(A)
        lam64, Q64 = lam.double().cpu(), Q.double().cpu()
        res = float((A32 @ Q64 - Q64 * lam64).abs().max() / A32.abs().max())
        orth = float((Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max())
        inv = Q64 @ ((Q64.T @ x) / (lam64 + d)[:, None])
        e_inv = float((inv - ref).abs().max() / ref.abs().max())
        print(f"n={n} rank {r} {name:18s}: residual {res:.1e} orth {orth:.1e} damped-inverse err {e_inv:.1e} min lam {float(lam64.min()):.1e}")
