"""Native eigensolver, one full-rank matrix: wall time of the phases of _eigh_native_group (synchronised in between)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip, eigh_native, linalg_native as L
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
X = torch.randn(2 * n, n, device="cuda"); A = X.T @ X / (2 * n)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T()
    An, scale = L._unit_scale(torch.stack([A]))
    ld = (n + 3) // 4 * 4
    work = torch.zeros(1, n, ld, device="cuda"); work[:, :, :n] = An
    t1 = T()
    D, E, tau = _hip.sytrd_(work[0], n)
    t2 = T()
    lam, Qt = eigh_native.stedc_native(D[None], E[None], n)
    t3 = T()
    Z = torch.zeros(1, n, ld, device="cuda"); Z[:, :, :n] = Qt.mT
    eigh_native.ormtr_native(work[0], tau, Z[0], n)
    t4 = T()
    Zc = Z[:, :, :n]
    G = _hip.gemm(Zc, Zc.mT); G.diagonal(dim1=-2, dim2=-1).sub_(1.0)
    R = _hip.gemm(An, Zc.mT) - Zc.mT * lam.unsqueeze(-2)
    ok = ((G.abs().amax(dim=(-2, -1)) <= 1e-3) & (R.abs().amax(dim=(-2, -1)) <= 1e-2)).tolist()
    t5 = T()
    print(f"n={n}: prep {1e3*(t1-t0):.1f}  sytrd {1e3*(t2-t1):.1f}  stedc {1e3*(t3-t2):.1f}  ormtr {1e3*(t4-t3):.1f}  verify {1e3*(t5-t4):.1f}  total {1e3*(t5-t0):.1f} ms  ok={ok}")
t0 = T(); L._eigh_full(A); t1 = T(); print(f"_eigh_full: {1e3*(t1-t0):.1f} ms")
