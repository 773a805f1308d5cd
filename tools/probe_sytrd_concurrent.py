"""k symmetric eigendecompositions of order n at once: own solver on k streams (one host thread each)
against torch.linalg.eigh stacked / on streams."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip, _rocsolver
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4609
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ld = (n + 3) // 4 * 4
mats = []
for i in range(k):
    X = torch.randn(513, n, device=dev)
    A0 = torch.zeros(n, ld, device=dev); A0[:, :n] = X.T @ X / 513
    mats.append(A0)

def own(A0):
    A = A0.clone()
    D, E, tau = _hip.sytrd_(A, n)
    Z = torch.empty(n, ld, device=dev)
    _rocsolver.stedc_(D, E, Z, n)
    _rocsolver.ormtr_(A, tau, Z, n)
    return D, Z

def run_threads(fn):
    streams = [torch.cuda.Stream() for _ in range(k)]
    def work(i):
        with torch.cuda.stream(streams[i]):
            fn(mats[i])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for th in ts: th.start()
    for th in ts: th.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3

for name, fn in (("own solver, streams", own), ("torch eigh, streams", lambda A: torch.linalg.eigh(A[:, :n]))):
    run_threads(fn)
    print(f"n={n} x{k}: {name}: {min(run_threads(fn) for _ in range(3)):.1f} ms", flush=True)
S = torch.stack([A[:, :n] for A in mats])
torch.linalg.eigh(S); torch.cuda.synchronize(); t = time.perf_counter(); torch.linalg.eigh(S); torch.cuda.synchronize()
print(f"n={n} x{k}: torch eigh stacked: {(time.perf_counter()-t)*1e3:.1f} ms")
torch.cuda.synchronize(); t = time.perf_counter(); own(mats[0]); torch.cuda.synchronize()
print(f"n={n} x1: own solver: {(time.perf_counter()-t)*1e3:.1f} ms")
