"""PSD matrices with many exactly-zero rows / columns (covariances of ReLU features with dead units): orthogonality
of the eigenvectors returned by torch.linalg.eigh (rocSOLVER) and by eigh_sytrd, float32 on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import linalg_native as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
for n, r, dead in ((845, 8, 0.0), (845, 8, 0.5), (845, 8, 0.9), (845, 16, 0.7), (300, 5, 0.8), (845, 845, 0.5), (2000, 12, 0.6)):
    X = torch.randn(r, n, dtype=torch.float64).clamp_min(0.0)        # ReLU features
    X[:, torch.rand(n) < dead] = 0.0                                  # dead units: exact zero columns
    A = (X.T @ X / r).float().to(dev)
    A32 = A.double().cpu()
    I = torch.eye(n, dtype=torch.float64)
    out = []
    for name, fn in (("torch raw", lambda M: tuple(torch.linalg.eigh(M))), ("torch normalised", L._torch_eigh_scaled), ("sytrd", L.eigh_sytrd),
                     ("torch fp64 on GPU", lambda M: tuple(torch.linalg.eigh(M.double())))):
        lam, Q = fn(A)
        Q64, l64 = Q.double().cpu(), lam.double().cpu()
        out.append(f"{name}: orth {float((Q64.T @ Q64 - I).abs().max()):.1e} res {float((A32 @ Q64 - Q64 * l64).abs().max() / A32.abs().max()):.1e}")
    print(f"n={n} rows {r} dead {dead}: " + " | ".join(out), flush=True)
