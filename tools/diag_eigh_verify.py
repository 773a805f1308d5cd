"""Scratch: orthogonality / residual defects of the native eigh route (before the float64 fallback decision)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip, eigh_native, linalg_native as L
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
for n in [int(a) for a in sys.argv[1:]] or [577, 1153, 2305]:
    for kind in ("lowrank", "verylow", "wishart"):
        r = max(16, n // 3) if kind == "lowrank" else (8 if kind == "verylow" else 2 * n)
        X = torch.rand(r, n, generator=g).to(dev); A = X.T @ X / r
        An, scale = L._unit_scale(A)
        ld = (n + 3) // 4 * 4
        work = torch.zeros(n, ld, device=dev); work[:, :n] = An
        t0 = time.perf_counter(); D, E, tau = _hip.sytrd_(work, n); torch.cuda.synchronize(); t1 = time.perf_counter()
        lam, Qt = eigh_native.stedc_native(D, E, n); torch.cuda.synchronize(); t2 = time.perf_counter()
        T = torch.diag(D.double()) + torch.diag(E[: n - 1].double(), 1) + torch.diag(E[: n - 1].double(), -1)
        rt = float((T @ Qt.double() - Qt.double() * lam.double()).abs().max()); ot = float((Qt.T.double() @ Qt.double() - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
        Z = torch.zeros(n, ld, device=dev); Z[:, :n] = Qt.T
        eigh_native.ormtr_native(work, tau, Z, n); torch.cuda.synchronize(); t3 = time.perf_counter()
        Q = Z[:, :n].T
        orth = float(L._orth_defect(Q)); res = float(L._residual_defect(An, lam, Q))
        nf = float(torch.linalg.matrix_norm(An)); n2 = float(lam.abs().max()); tol = float(L._residual_tol(An))
        print(f"n={n:5d} {kind:8s} sytrd {1e3*(t1-t0):7.1f} stedc {1e3*(t2-t1):7.1f} ormtr {1e3*(t3-t2):6.1f} ms | tridiag: res {rt:.1e} orth {ot:.1e} | full: orth {orth:.1e} (tol {L._ORTH_TOL}) res {res:.1e} = {res/(L._EPS32*nf):.1f} eps|A|_F = {res/(L._EPS32*n2):.1f} eps|A|_2 (|A|_F {nf:.3g}, |A|_2 {n2:.3g}; tol {tol:.1e})", flush=True)
