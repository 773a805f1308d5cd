import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from curvlinops_amd import _hip, linalg_native as L
torch.manual_seed(0)
n = 4608
def T():
    torch.cuda.synchronize(); return time.perf_counter()
def run(name, A):
    ld = n
    for _ in range(2):
        work = torch.zeros(n, ld, device="cuda"); work[:, :n] = A
        t1 = T(); _hip.sytrd_(work, n); t2 = T()
    print(f"{name}: sytrd {1e3*(t2-t1):.1f} ms; |A|max {float(A.abs().max()):.3g} min diag {float(A.diagonal().min()):.3g}")
X = torch.randn(n, n, device="cuda"); run("X X^T / n (square X)", X @ X.T / n)
X2 = torch.randn(2 * n, n, device="cuda"); A2 = X2.T @ X2 / (2 * n); run("X^T X / 2n", A2)
run("same, unit scale", A2 / A2.abs().max())
run("same x 1e-3", A2 * 1e-3)
run("identity + small", torch.eye(n, device="cuda") + 0.01 * (A2 / A2.abs().max()))
