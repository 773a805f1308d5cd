"""Host (Python glue) time vs device time of the native eigensolver stages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip, eigh_native, linalg_native
_hip.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for n in (333, 577, 1153, 2305, 4609):
    X = torch.rand(max(16, n // 3), n, generator=g).to(dev)
    A = X.T @ X / X.shape[0]
    An = A / A.abs().max()
    ld = (n + 3) // 4 * 4
    def stage():
        work = torch.zeros(n, ld, device=dev); work[:, :n] = An
        torch.cuda.synchronize(); t = [time.perf_counter()]
        D, E, tau = _hip.sytrd_(work, n); t.append(time.perf_counter()); torch.cuda.synchronize(); t.append(time.perf_counter())
        lam, Qt = eigh_native.stedc_native(D, E, n); t.append(time.perf_counter()); torch.cuda.synchronize(); t.append(time.perf_counter())
        Z = torch.zeros(n, ld, device=dev); Z[:, :n] = Qt.T
        torch.cuda.synchronize(); t.append(time.perf_counter())
        eigh_native.ormtr_native(work, tau, Z, n); t.append(time.perf_counter()); torch.cuda.synchronize(); t.append(time.perf_counter())
        return t
    stage(); t = stage()
    ms = lambda a, b: 1e3 * (t[b] - t[a])
    print(f"n={n:5d} sytrd host {ms(0,1):6.2f} total {ms(0,2):7.2f} | stedc host {ms(2,3):6.2f} total {ms(2,4):6.2f} | ormtr host {ms(5,6):6.2f} total {ms(5,7):6.2f} ms")
