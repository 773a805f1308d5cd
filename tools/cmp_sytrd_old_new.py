"""Round 4 scratch: clo_sytrd_f32 (persistent panel launches) against the round-3 column-launch kernel built into one
comparison library (libclo_sycmp.so): D, E, tau, reflectors, and each against float64 (Q^T A Q = T with the STORED v, tau)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "curvlinops_amd", "lib", "variants", "libclo_sycmp.so"))
P, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
lib.clo_sytrd_f32.argtypes = [P, L, I, P, P, P, P, L, I, P]
lib.clo_sytrd_old_f32.argtypes = [P, L, I, P, P, P, P, L, P]
lib.clo_sytrd_ws_bytes.restype = L; lib.clo_sytrd_old_ws_bytes.restype = L
lib.clo_sytrd_ws_bytes.argtypes = [I]; lib.clo_sytrd_old_ws_bytes.argtypes = [I]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def run(which, A, n, maxb=0):
    ld = (n + 3) // 4 * 4
    work = torch.zeros(n, ld, device=dev); work[:, :n] = A
    D, E, tau = (torch.zeros(n, device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    if which == "new":
        nb = lib.clo_sytrd_ws_bytes(n); ws = torch.zeros(nb // 4, device=dev)
        rc = lib.clo_sytrd_f32(work.data_ptr(), ld, n, D.data_ptr(), E.data_ptr(), tau.data_ptr(), ws.data_ptr(), nb, maxb, st)
    else:
        nb = lib.clo_sytrd_old_ws_bytes(n); ws = torch.zeros(nb // 4, device=dev)
        rc = lib.clo_sytrd_old_f32(work.data_ptr(), ld, n, D.data_ptr(), E.data_ptr(), tau.data_ptr(), ws.data_ptr(), nb, st)
    torch.cuda.synchronize()
    assert rc == 0
    return D.cpu().double(), E.cpu().double(), tau.cpu().double(), work[:, :n].cpu().double()

def quality(A64, D, E, tau, work, n):
    """|Q^T A Q - T| / |A| with Q = H_0 ... H_{n-3} from the stored reflectors (float64 on the host)."""
    A64, D, E, tau, work = (t.to(dev) for t in (A64, D, E, tau, work))
    Q = torch.eye(n, dtype=torch.float64, device=dev)
    for j in range(n - 2):
        v = torch.zeros(n, dtype=torch.float64, device=dev); v[j + 1] = 1.0; v[j + 2:] = work[j, j + 2:]
        Q = Q - tau[j] * torch.outer(Q @ v, v)
    T = torch.diag(D) + torch.diag(E[: n - 1], 1) + torch.diag(E[: n - 1], -1)
    return (float((Q.T @ A64 @ Q - T).abs().max() / A64.abs().max()),
            float((Q.T @ Q - torch.eye(n, dtype=torch.float64, device=dev)).abs().max()))

for n in (577, 1153, 2305):
    for kind in ("lowrank", "wishart"):
        r = max(16, n // 3) if kind == "lowrank" else 2 * n
        X = torch.rand(r, n, generator=g).to(dev)
        A = X.T @ X / r; A = A / A.abs().max()
        A64 = A.cpu().double()
        res = {}
        for which, mb in (("old", 0), ("new", 0), ("new", 72)):
            D, E, tau, work = run(which, A, n, mb)
            q = quality(A64, D, E, tau, work, n)
            res[(which, mb)] = (D, E, tau, work)
            print(f"n={n:5d} {kind:8s} {which}{mb:3d}: |Q^TAQ-T|/|A| {q[0]:.2e}  |Q^TQ-I| {q[1]:.2e}", flush=True)
        o, nw = res[("old", 0)], res[("new", 0)]
        print(f"     new vs old: D {float((o[0]-nw[0]).abs().max()):.2e} E {float((o[1]-nw[1]).abs().max()):.2e} "
              f"tau {float((o[2]-nw[2]).abs().max()):.2e} reflectors {float((torch.tril(o[3]-nw[3], -0)).abs().max()):.2e}"
              f" upper(v) {float((torch.triu(o[3]-nw[3], 2)).abs().max()):.2e}")
