"""Randomised check of the damped Cholesky inverse (clo_cholesky_inverse_f32 behind
linalg_native.damped_cholesky_inverse): random orders (odd ones, tiny ones), scales 1e-6 ... 1e6, condition
numbers, dampings, against float64 LAPACK: |(A + d I) X - I| and |X - X_ref| relative.
    python tools/fuzz_cholesky.py [seed] [cases]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from curvlinops_amd.linalg_native import damped_cholesky_inverse


def run(seed, ncase):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst, fails = 0.0, []
    for case in range(ncase):
        n = int(rng.choice([1, 2, 3, 5, 17, 31, 32, 33, 64, 65, 127, 128, 129, 200, 257, 400, 577, 1000]))
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        cond = 10.0 ** rng.uniform(0, 4)
        lam = np.logspace(0, -np.log10(cond), n)
        p10 = int(rng.integers(-6, 7))
        A64 = (Q * lam) @ Q.T * 10.0 ** p10
        A64 = 0.5 * (A64 + A64.T)
        damping = float(10.0 ** p10 * 10.0 ** rng.uniform(-4, 0)) if rng.random() < 0.7 else 0.0
        A = torch.as_tensor(A64, dtype=torch.float32, device=dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            X = damped_cholesky_inverse(A, damping)
        A32 = A.double().cpu().numpy() + damping * np.eye(n)
        Xref = np.linalg.inv(A32)
        X64 = X.double().cpu().numpy()
        kappa = np.linalg.cond(A32)
        err = np.abs(X64 - Xref).max() / np.abs(Xref).max()
        res = np.abs(A32 @ X64 - np.eye(n)).max()
        tol = 3e-6 * kappa + 1e-5
        worst = max(worst, err / tol)
        if not (err < tol and np.isfinite(X64).all()):
            fails.append(f"case {case}: n={n} scale 1e{p10} cond {kappa:.1e} damping {damping:.1e}: err {err:.1e} (tol {tol:.1e}) residual {res:.1e}")
    return worst, fails


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    worst, fails = run(seed, ncase)
    for f in fails:
        print(f)
    print(f"done: {ncase} cases, worst err / tolerance {worst:.2f}, {len(fails)} failures")
