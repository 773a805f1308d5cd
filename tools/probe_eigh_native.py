"""Accuracy / time of the hand-written eigensolver tail (divide & conquer + block-reflector back-transformation)
against rocSOLVER's sstedc / sormtr behind the same own reduction, and of stedc_native alone on tridiagonal
matrices against float64 LAPACK."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from curvlinops_amd import _hip, eigh_native, linalg_native

_hip.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def tri_case(kind, n):
    if kind == "random":
        d, e = torch.rand(n, generator=g) - 0.5, torch.rand(n, generator=g) - 0.5
    elif kind == "toeplitz":   # -1 2 -1: well separated, known spectrum
        d, e = torch.full((n,), 2.0), torch.full((n,), -1.0)
    elif kind == "clustered":  # tiny couplings: many (nearly) equal eigenvalues
        d = torch.cat([torch.ones(n // 2), torch.rand(n - n // 2, generator=g)])
        e = torch.rand(n, generator=g) * 1e-6
    elif kind == "lowrank":    # tridiagonal form of a rank-deficient PSD matrix
        X = torch.rand(max(8, n // 6), n, generator=g, dtype=torch.float64)
        A = (X.T @ X).to(dev).float()
        ld = (n + 3) // 4 * 4
        work = torch.zeros(n, ld, device=dev); work[:, :n] = A / A.abs().max()
        D, E, _ = _hip.sytrd_(work, n)
        return D.cpu(), E.cpu()
    return d, e


print("stedc_native on tridiagonal matrices: max |lam - lam64| / |lam|max, |Q^T Q - I|, |T Q - Q lam| / |T|, ms")
for kind in ("random", "toeplitz", "clustered", "lowrank"):
    for n in (1, 2, 5, 64, 65, 200, 577, 1153, 2305, 4609):
        if kind == "lowrank" and n < 64:
            continue
        d, e = tri_case(kind, n)
        T = torch.diag(d.double()) + torch.diag(e[: n - 1].double(), 1) + torch.diag(e[: n - 1].double(), -1)
        ref = np.linalg.eigvalsh(T.numpy())
        dd, ee = d.to(dev).float(), e.to(dev).float()
        eigh_native.stedc_native(dd, ee, n); torch.cuda.synchronize()
        t0 = time.perf_counter()
        lam, Q = eigh_native.stedc_native(dd, ee, n)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        lam64, Q64 = lam.double().cpu(), Q.double().cpu()
        scale = max(np.abs(ref).max(), 1e-30)
        err = np.abs(lam64.numpy() - ref).max() / scale
        orth = (Q64.T @ Q64 - torch.eye(n, dtype=torch.float64)).abs().max().item()
        res = (T @ Q64 - Q64 * lam64[None, :]).abs().max().item() / max(T.abs().max().item(), 1e-30)
        print(f"{kind:10s} n={n:5d}  lam {err:.2e}  orth {orth:.2e}  res {res:.2e}  {ms:8.2f} ms")

print("full eigh through the own reduction: native tail vs rocSOLVER tail (orth, residual on the normalised matrix, ms)")
for n in (64, 333, 577, 1153, 2305, 4609):
    X = torch.rand(max(16, n // 3), n, generator=g).to(dev)
    A = X.T @ X / X.shape[0]
    for tail in ("native", "rocsolver"):
        linalg_native._EIGH_VENDOR_TAIL = tail == "rocsolver"
        linalg_native.eigh_sytrd(A); torch.cuda.synchronize()
        t0 = time.perf_counter()
        lam, Q = linalg_native.eigh_sytrd(A)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        An = A / A.abs().max()
        orth = float(linalg_native._orth_defect(Q))
        res = float(linalg_native._residual_defect(An, lam / A.abs().max(), Q))
        rec = float((Q * lam[None, :] @ Q.T - A).abs().max() / A.abs().max())
        print(f"n={n:5d} {tail:9s} orth {orth:.2e}  res {res:.2e}  |Q lam Q^T - A| {rec:.2e}  {ms:8.2f} ms")
