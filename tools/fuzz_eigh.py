"""Randomised check of eigh_sytrd (clo_sytrd_f32 -> sstedc -> sormtr): random orders, spectra (well separated,
clustered, rank deficient, indefinite, scaled by 1e-6 ... 1e6, identity, rank one, block diagonal) against float64
LAPACK: eigenvalues, residual and orthogonality relative to |A|.    python tools/fuzz_eigh.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from curvlinops_amd import linalg_native as L
from curvlinops_amd.linalg_native import eigh_sytrd


def make(rng, n, kind):
    if kind == "spectrum":
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        style = rng.integers(0, 4)
        if style == 0:
            lam = rng.standard_normal(n)
        elif style == 1:   # clusters
            lam = np.resize(np.repeat(rng.standard_normal(max(1, n // 8)), 8), n) + 1e-7 * rng.standard_normal(n)
        elif style == 2:   # geometric decay, PSD
            lam = np.logspace(0, -8, n)
        else:              # rank deficient
            lam = np.concatenate([rng.random(max(1, n // 6)) + 0.1, np.zeros(n - max(1, n // 6))])
        return (Q * lam) @ Q.T
    if kind == "gram":
        r = int(rng.integers(1, n + 3))
        X = rng.standard_normal((r, n)) * np.logspace(0, -rng.integers(0, 5), n)
        return X.T @ X / r
    if kind == "identity":
        return np.eye(n) * rng.standard_normal()
    if kind == "rank1":
        v = rng.standard_normal(n)
        return np.outer(v, v)
    if kind == "blockdiag":
        A = np.zeros((n, n)); h = n // 2
        B = rng.standard_normal((h, h)); A[:h, :h] = B + B.T
        Cc = rng.standard_normal((n - h, n - h)); A[h:, h:] = Cc @ Cc.T
        return A
    raise ValueError(kind)


def run(seed, ncase):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst = [0.0, 0.0, 0.0]
    fails = []
    for case in range(ncase):
        n = int(rng.choice([3, 4, 5, 9, 17, 33, 64, 65, 100, 129, 200, 257, 400, 513, 777]))
        kind = str(rng.choice(["spectrum", "gram", "identity", "rank1", "blockdiag"], p=[0.45, 0.35, 0.05, 0.05, 0.10]))
        p10 = int(rng.integers(-6, 7)) if os.environ.get("CLO_FUZZ_EIGH_NOSCALE") is None else 0
        A64 = make(rng, n, kind) * 10.0 ** p10
        A64 = 0.5 * (A64 + A64.T)
        A = torch.as_tensor(A64, dtype=torch.float32, device=dev)
        lam, Q = eigh_sytrd(A) if os.environ.get("CLO_FUZZ_EIGH_DEFAULT") is None else L.eigh(A)
        A32 = A.double().cpu().numpy()   # what the solver saw
        ref = np.linalg.eigvalsh(A32)
        sc = max(np.abs(A32).max(), 1e-300)
        l64, Q64 = lam.double().cpu().numpy(), Q.double().cpu().numpy()
        e_val = np.abs(l64 - ref).max() / max(np.abs(ref).max(), 1e-300)
        e_res = np.abs(A32 @ Q64 - Q64 * l64).max() / sc
        e_orth = np.abs(Q64.T @ Q64 - np.eye(n)).max()
        worst = [max(worst[0], e_val), max(worst[1], e_res), max(worst[2], e_orth)]
        if not (e_val < 2e-5 and e_res < 2e-5 * max(1.0, n ** 0.5 / 8) and e_orth < 2e-5 and np.isfinite(l64).all()):
            fails.append(f"case {case}: n={n} {kind} scale 1e{p10}: eigenvalues {e_val:.1e} residual {e_res:.1e} orth {e_orth:.1e}")
    return worst, fails


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    worst, fails = run(seed, ncase)
    for f in fails:
        print(f)
    print(f"done: {ncase} cases, worst eigenvalues {worst[0]:.1e} residual {worst[1]:.1e} orth {worst[2]:.1e}, {len(fails)} failures")
