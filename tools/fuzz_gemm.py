"""Randomised check of clo_gemm_f32 / clo_syrk_accum_f32 (all engines: aligned v2 tiles, 64x64x64
small tiles, v1, the one-launch tiny kernel; transposed views, batches, alpha / beta, split-K)
against float64 torch.    python tools/fuzz_gemm.py [seed] [cases]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from curvlinops_amd import _hip


def run(seed: int, ncase: int):
    _hip.load()
    rng = np.random.default_rng(seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    worst, failures = 0.0, []
    for case in range(ncase):
        kind = rng.integers(0, 4)
        def dim():
            r = rng.random()
            if r < 0.4: return int(rng.integers(1, 70))
            if r < 0.8: return int(rng.integers(1, 40)) * 4
            return int(rng.integers(100, 700))
        M, N, K = dim(), dim(), dim()
        nb = int(rng.choice([1, 1, 1, 2, 3]))
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        alpha = float(rng.choice([1.0, -0.5, 2.0])); beta = float(rng.choice([0.0, 0.0, 1.0, 0.5]))
        shapeA = (nb, K, M) if ta else (nb, M, K)
        shapeB = (nb, N, K) if tb else (nb, K, N)
        A = torch.rand(*shapeA, device="cuda", generator=g) - 0.5
        B = torch.rand(*shapeB, device="cuda", generator=g) - 0.5
        Av = A.transpose(1, 2) if ta else A
        Bv = B.transpose(1, 2) if tb else B
        if nb == 1 and rng.random() < 0.5:
            Av, Bv = Av[0], Bv[0]
        C0 = torch.rand(*( (nb, M, N) if Av.dim() == 3 else (M, N)), device="cuda", generator=g)
        out = C0.clone()
        splitk = None if rng.random() < 0.6 else int(rng.integers(1, 5))
        what = f"case {case}: M={M} N={N} K={K} nb={nb} ta={ta} tb={tb} alpha={alpha} beta={beta} splitk={splitk}"
        try:
            if kind == 3 and Av.dim() == 2:  # SYRK of a random tall matrix (+ ones column)
                ones = bool(rng.integers(0, 2))
                X = torch.rand(K, M, device="cuda", generator=g) - 0.5
                d = M + (1 if ones else 0)
                Cs = torch.rand(d, d, device="cuda", generator=g); Cs = Cs + Cs.T
                ref = beta * Cs.double()
                Xe = torch.cat([X, torch.ones(K, 1, device="cuda")], 1) if ones else X
                ref = ref + alpha * (Xe.double().T @ Xe.double())
                got = _hip.syrk_accum(Cs.clone(), X, alpha=alpha, beta=beta, ones_col=ones)
                what = f"case {case}: SYRK rows={K} d={M} ones={ones} alpha={alpha} beta={beta}"
            else:
                got = _hip.gemm(Av, Bv, out=out, alpha=alpha, beta=beta, splitk=splitk)
                ref = alpha * (Av.double() @ Bv.double()) + beta * C0.double()
        except Exception as e:  # noqa: BLE001
            failures.append(f"exception in {what}: {e}")
            continue
        err = float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        worst = max(worst, err)
        if not err < 2e-5:
            failures.append(f"mismatch in {what}: err={err:.3e}")
    return worst, failures


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    worst, failures = run(seed, ncase)
    for f in failures[:20]:
        print(f)
    print(f"done: {ncase} cases, worst rel err {worst:.2e}, {len(failures)} failures")
