"""Where does rocSOLVER's symmetric eigensolver spend its time?  Times ssytrd / sstedc / sormtr
separately through the rocSOLVER C API (ctypes) next to torch.linalg.eigh, on KFAC-like PSD matrices.

    python tools/probe_rocsolver_phases.py [n ...]
"""
import ctypes
import os
import sys
import time

import torch

lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
rocblas = ctypes.CDLL(os.path.join(lib_dir, "librocblas.so"), mode=ctypes.RTLD_GLOBAL)
rs = ctypes.CDLL(os.path.join(lib_dir, "librocsolver.so"), mode=ctypes.RTLD_GLOBAL)
P, I = ctypes.c_void_p, ctypes.c_int
handle = P()
assert rocblas.rocblas_create_handle(ctypes.byref(handle)) == 0
rocblas.rocblas_set_stream.argtypes = [P, P]
rs.rocsolver_ssytrd.argtypes = [P, I, I, P, I, P, P, P]
rs.rocsolver_sstedc.argtypes = [P, I, I, P, P, P, I, P]
rs.rocsolver_sormtr.argtypes = [P, I, I, I, I, I, P, I, P, P, I]
LOWER, UPPER, LEFT, NONE, EV_TRI = 122, 121, 141, 111, 212


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3


def main(ns):
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    rocblas.rocblas_set_stream(handle, P(torch.cuda.current_stream().cuda_stream))
    for n in ns:
        g = torch.Generator(device=dev).manual_seed(n)
        rows = min(n, 513)  # KFAC factor of a 512-row batch: rank-deficient PSD
        X = torch.randn(rows, n, device=dev, generator=g) * torch.logspace(0, -3, n, device=dev)
        A0 = (X.T @ X) / rows
        t_eigh = timed(lambda: torch.linalg.eigh(A0))
        A = A0.clone()
        D = torch.empty(n, device=dev)
        E = torch.empty(n, device=dev)
        tau = torch.empty(n, device=dev)
        Z = torch.empty(n, n, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)

        def sytrd():
            A.copy_(A0)
            assert rs.rocsolver_ssytrd(handle, LOWER, n, A.data_ptr(), n, D.data_ptr(), E.data_ptr(), tau.data_ptr()) == 0

        t_copy = timed(lambda: A.copy_(A0))
        t_sytrd = timed(sytrd) - t_copy
        d0, e0 = D.clone(), E.clone()

        def stedc():
            D.copy_(d0)
            E.copy_(e0)
            assert rs.rocsolver_sstedc(handle, EV_TRI, n, D.data_ptr(), E.data_ptr(), Z.data_ptr(), n, info.data_ptr()) == 0

        t_stedc = timed(stedc)

        def ormtr():
            assert rs.rocsolver_sormtr(handle, LEFT, LOWER, NONE, n, n, A.data_ptr(), n, tau.data_ptr(), Z.data_ptr(), n) == 0

        t_ormtr = timed(ormtr, reps=1)
        # check: column-major Z holds eigenvectors as columns  ->  torch sees Z^T
        Q = Z.T
        lam = D
        res = float((A0 @ Q - Q * lam).abs().max() / A0.abs().max())
        orth = float((Q.T @ Q - torch.eye(n, device=dev)).abs().max())
        print(f"n={n:5d}: eigh {t_eigh:7.1f} ms | sytrd {t_sytrd:7.1f}  stedc {t_stedc:7.1f}  ormtr {t_ormtr:6.1f} ms"
              f" | residual {res:.1e} orth {orth:.1e} info {int(info)}", flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [577, 1153, 2305, 4609])
