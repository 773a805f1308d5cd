"""Re-create one case of tools/fuzz_eigh.py and check every stage: python tools/diag_eigh_case.py seed case"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fuzz_eigh as F
from curvlinops_amd import _hip, linalg_native as L
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
for case in range(target + 1):
    n = int(rng.choice([3, 4, 5, 9, 17, 33, 64, 65, 100, 129, 200, 257, 400, 513, 777]))
    kind = str(rng.choice(["spectrum", "gram", "identity", "rank1", "blockdiag"], p=[0.45, 0.35, 0.05, 0.05, 0.10]))
    p10 = int(rng.integers(-6, 7))
    A64 = F.make(rng, n, kind) * 10.0 ** p10
A64 = 0.5 * (A64 + A64.T)
A = torch.as_tensor(A64, dtype=torch.float32, device=dev)
An, s = L._unit_scale(A)
A32 = An.double().cpu().numpy()
ref = np.linalg.eigvalsh(A32)
print(f"n={n} {kind} scale 1e{p10}; spectrum of A/|A|: max {ref.max():.3e} min {ref.min():.3e}, #|lam|<1e-6: {(np.abs(ref)<1e-6).sum()}")
ld = (n + 3) // 4 * 4
for mode in ("fast+careful", "always careful"):
    if mode == "always careful":
        os.environ["CLO_TD_CAREFUL_ALL"] = "1"
    W = torch.zeros(n, ld, device=dev); W[:, :n] = An
    D, E, tau = _hip.sytrd_(W, n)
    T = np.diag(D.double().cpu().numpy()) + np.diag(E[:n-1].double().cpu().numpy(), 1) + np.diag(E[:n-1].double().cpu().numpy(), -1)
    print(f"  own sytrd ({mode}): tridiagonal spectrum err {np.abs(np.linalg.eigvalsh(T) - ref).max():.2e}")
lt = torch.linalg.eigvalsh(An).double().cpu().numpy()
print(f"  torch.linalg.eigvalsh on the normalised matrix: err {np.abs(lt - ref).max():.2e}")
lam, Q = L.eigh_sytrd(A)
print(f"  eigh_sytrd eigenvalue err {np.abs(lam.double().cpu().numpy()/float(s) - ref).max():.2e}")
