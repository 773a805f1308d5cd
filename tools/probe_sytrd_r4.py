"""Round 4: time of clo_sytrd_f32 (persistent panel launches) and of the full native eigh on PSD factors of the benchmark
sizes; checks the tridiagonal spectrum against float64 LAPACK.  MAXB=<n> caps the workgroups per panel launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from curvlinops_amd import _hip, linalg_native

_hip.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
maxb = int(os.environ.get("MAXB", "0"))
for n in [int(a) for a in sys.argv[1:]] or [577, 1153, 2305, 4609]:
    for kind in ("lowrank", "wishart"):
        r = max(16, n // 3) if kind == "lowrank" else 2 * n
        X = torch.rand(r, n, generator=g).to(dev)
        A = X.T @ X / r
        A = A / A.abs().max()
        ld = (n + 3) // 4 * 4
        ts = []
        for rep in range(3):
            work = torch.zeros(n, ld, device=dev); work[:, :n] = A
            torch.cuda.synchronize(); t0 = time.perf_counter()
            D, E, tau = _hip.sytrd_(work, n, max_blocks=maxb)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        d, e = D.double().cpu().numpy(), E.double().cpu().numpy()[: n - 1]
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        ref = np.linalg.eigvalsh(A.double().cpu().numpy())
        err = np.abs(np.linalg.eigvalsh(T) - ref).max() / np.abs(ref).max()
        te = []
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lam, Q = linalg_native.eigh(A)
            torch.cuda.synchronize(); te.append(1e3 * (time.perf_counter() - t0))
        rec = ((Q * lam) @ Q.T - A).abs().max().item()
        orth = (Q.T @ Q - torch.eye(n, device=dev)).abs().max().item()
        print(f"n={n:5d} {kind:8s} sytrd {min(ts):8.2f} ms ({1e3*min(ts)/n:5.1f} us/col)  spectrum err {err:.1e}   "
              f"eigh {min(te):8.2f} ms  |QLQ^T-A| {rec:.1e}  |Q^TQ-I| {orth:.1e}", flush=True)
