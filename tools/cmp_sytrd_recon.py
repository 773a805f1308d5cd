"""Scratch: reconstruction error of the full native eigh at n = 4609 (the test's matrix) with the reduction done by the
round-3 column kernel / the persistent panel kernel at several workgroup caps (comparison library libclo_sycmp.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import eigh_native, linalg_native as L
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "curvlinops_amd", "lib", "variants", "libclo_sycmp.so"))
P, Lg, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
lib.clo_sytrd_f32.argtypes = [P, Lg, I, P, P, P, P, Lg, I, P]
lib.clo_sytrd_old_f32.argtypes = [P, Lg, I, P, P, P, P, Lg, P]
lib.clo_sytrd_ws_bytes.restype = Lg; lib.clo_sytrd_old_ws_bytes.restype = Lg
lib.clo_sytrd_ws_bytes.argtypes = [I]; lib.clo_sytrd_old_ws_bytes.argtypes = [I]
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [4609]:
    g = torch.Generator().manual_seed(n)
    X = torch.rand(max(16, n // 3), n, generator=g, dtype=torch.float64)
    A64 = X.T @ X / X.shape[0]
    A = A64.to(dev, torch.float32)
    An, scale = L._unit_scale(A)
    ld = (n + 3) // 4 * 4
    for which, mb in (("old", 0), ("new", 128), ("new", 96), ("new", 64), ("new", 32)):
        work = torch.zeros(n, ld, device=dev); work[:, :n] = An
        D, E, tau = (torch.zeros(n, device=dev) for _ in range(3))
        st = torch.cuda.current_stream().cuda_stream
        if which == "new":
            nb = lib.clo_sytrd_ws_bytes(n); ws = torch.zeros(nb // 4, device=dev)
            rc = lib.clo_sytrd_f32(work.data_ptr(), ld, n, D.data_ptr(), E.data_ptr(), tau.data_ptr(), ws.data_ptr(), nb, mb, st)
        else:
            nb = lib.clo_sytrd_old_ws_bytes(n); ws = torch.zeros(nb // 4, device=dev)
            rc = lib.clo_sytrd_old_f32(work.data_ptr(), ld, n, D.data_ptr(), E.data_ptr(), tau.data_ptr(), ws.data_ptr(), nb, st)
        torch.cuda.synchronize(); assert rc == 0
        lam, Qt = eigh_native.stedc_native(D, E, n)
        Z = torch.zeros(n, ld, device=dev); Z[:, :n] = Qt.T
        eigh_native.ormtr_native(work, tau, Z, n)
        Q = Z[:, :n].T
        lam = lam * scale.reshape(())
        Qd, ld_ = Q.double(), lam.double()
        rec = float(((Qd * ld_) @ Qd.T - A.double()).abs().max()) / float(A64.abs().max())
        orth = float((Qd.T @ Qd - torch.eye(n, dtype=torch.float64, device=dev)).abs().max())
        print(f"n={n} {which:3s} G<={mb:3d}: |Q L Q^T - A| / |A|max {rec:.2e}   |Q^T Q - I| {orth:.1e}", flush=True)
