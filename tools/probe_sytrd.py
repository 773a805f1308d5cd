"""clo_sytrd_f32 against rocSOLVER's ssytrd (same LAPACK conventions -> same d, e, tau up to rounding),
and the full eigensolver  sytrd -> sstedc -> sormtr  against torch.linalg.eigh: residual, orthogonality,
time.    python tools/probe_sytrd.py [n ...]
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from curvlinops_amd import _hip
import _rocsolver  # tools/_rocsolver.py

P, I = ctypes.c_void_p, ctypes.c_int


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3


def padded(A0):
    n = A0.shape[0]
    ld = (n + 3) // 4 * 4
    A = torch.zeros(n, ld, device=A0.device)
    A[:, :n] = A0
    return A


def main(ns):
    dev = torch.device("cuda:0")
    rb, rs = _rocsolver._load()
    rs.rocsolver_ssytrd.argtypes = [P, I, I, P, I, P, P, P]
    for n in ns:
        g = torch.Generator(device=dev).manual_seed(n)
        rows = min(n, 513)
        X = torch.randn(rows, n, device=dev, generator=g) * torch.logspace(0, -3, n, device=dev)
        A0 = (X.T @ X) / rows
        A0 = 0.5 * (A0 + A0.T)
        # reference reduction
        Ar = padded(A0)
        Dr = torch.empty(n, device=dev); Er = torch.empty(n, device=dev); taur = torch.empty(n, device=dev)
        h = _rocsolver._handle(dev)
        assert rs.rocsolver_ssytrd(h, 122, n, Ar.data_ptr(), Ar.stride(0), Dr.data_ptr(), Er.data_ptr(), taur.data_ptr()) == 0
        A = padded(A0)
        D, E, tau = _hip.sytrd_(A, n)
        torch.cuda.synchronize()
        sc = float(A0.abs().max())
        dd = float((D - Dr).abs().max()) / sc
        de = float((E[: n - 1] - Er[: n - 1]).abs().max()) / sc
        dt = float((tau[: n - 2] - taur[: n - 2]).abs().max())
        dv = float((torch.tril(A[:, :n].T, -2) - torch.tril(Ar[:, :n].T, -2)).abs().max())
        # full solver
        def solve():
            A = padded(A0)
            D, E, tau = _hip.sytrd_(A, n)
            Z = torch.empty(n, A.stride(0), device=dev)
            info = _rocsolver.stedc_(D, E, Z, n)
            _rocsolver.ormtr_(A, tau, Z, n)
            return D, Z[:, :n].T, info
        lam, Q, info = solve()
        res = float((A0 @ Q - Q * lam).abs().max() / sc)
        orth = float((Q.T @ Q - torch.eye(n, device=dev)).abs().max())
        ref = torch.linalg.eigvalsh(A0.double())
        dl = float((lam.double() - ref).abs().max() / ref.abs().max())
        t_own = timed(solve)
        t_red = timed(lambda: _hip.sytrd_(padded(A0), n))
        t_ref = timed(lambda: torch.linalg.eigh(A0))
        print(f"n={n:5d}: d {dd:.1e} e {de:.1e} tau {dt:.1e} v {dv:.1e} | residual {res:.1e} orth {orth:.1e} "
              f"eigenvalues {dl:.1e} info {int(info)} | sytrd {t_red:6.1f} ms, eigh own {t_own:6.1f} vs torch {t_ref:6.1f} ms",
              flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [5, 66, 130, 577, 1153, 2305, 4609])
