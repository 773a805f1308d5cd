"""Round 4: persistent grids of different streams side by side (csrc/persist_gate.h): sytrd panel launches on one or two
streams next to persistent GGN products on another.  MODE = ss (two reductions), sm (reduction + products), ssm (all)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from curvlinops_amd import _hip
_hip.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "ssm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ld = (n + 3) // 4 * 4
mats = []
for r in (n // 3, 2 * n):
    X = torch.rand(r, n, generator=g).to(dev); A = X.T @ X / r; mats.append(A / A.abs().max())
def reduce(A):
    P = torch.zeros(n, ld, device=dev); P[:, :n] = A
    return _hip.sytrd_(P, n)
serial = [reduce(A) for A in mats]
dims, acts = [1024, 2688, 2688, 10], [1, 1, 0]
W = [torch.randn(dims[i + 1], dims[i], device=dev) / dims[i] ** 0.5 for i in range(3)]
b = [torch.randn(dims[i + 1], device=dev) * 0.1 for i in range(3)]
VW = [torch.rand_like(w) for w in W]; Vb = [torch.rand_like(x) for x in b]
plan = _hip.MLPPlan(dims, acts)
X8 = torch.rand(8, dims[0], device=dev)
def product():
    OW = [torch.empty_like(w) for w in W]; Ob = [torch.empty_like(x) for x in b]
    plan.ggn_matvec(W, b, VW, Vb, OW, Ob, X8, 0, 2.0 / 80, 1.0, 0.0)
    return OW + Ob
ref = product(); torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(3)]
t0 = time.perf_counter()
red, prods = [[], []], []
for rep in range(3):
    for i in range(2 if "ss" in mode else 1):
        if "s" in mode:
            with torch.cuda.stream(streams[i]):
                red[i].append(reduce(mats[i]))
    if "m" in mode:
        with torch.cuda.stream(streams[2]):
            for _ in range(10):
                prods.append(product())
    print("rep", rep, "queued", flush=True)
torch.cuda.synchronize()
print(f"mode {mode}: done in {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for i in range(2):
    for D, E, tau in red[i]:
        from scipy.linalg import eigvalsh_tridiagonal as _ev
        lam = _ev(D[:n].double().cpu().numpy(), E[: n - 1].double().cpu().numpy())
        lam0 = _ev(serial[i][0][:n].double().cpu().numpy(), serial[i][1][: n - 1].double().cpu().numpy())
        assert abs(lam - lam0).max() <= 1e-4 * abs(lam0).max(), "reduction differs"
for out in prods:
    assert all(torch.equal(a, c) for a, c in zip(out, ref)), "product differs"
print("results equal the serial ones")
