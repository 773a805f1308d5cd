"""Scratch: correctness of the LDS-DMA GEMM engine (one tile per workgroup, split-K slabs, stream-K) vs float64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
lib = _hip.load()
torch.manual_seed(0)
bad = 0
cases = []
for (M, N, K) in ((512, 2304, 2304), (256, 2304, 2304), (1024, 1024, 1024), (516, 1156, 1000), (132, 128, 4096),
                  (128, 128, 2304), (2048, 2048, 512), (640, 640, 36), (4608, 512, 4608), (300, 3000, 68)):
    for ta in (False, True):
        for tb in (False, True):
            for sk in (None, 1, 3):
                cases.append((M, N, K, ta, tb, sk, 1))
cases += [(256, 384, 512, False, False, None, 5), (256, 384, 512, True, True, None, 3), (512, 512, 260, False, True, 2, 2)]
for (M, N, K, ta, tb, sk, nb) in cases:
    A = torch.randn((nb, K, M) if ta else (nb, M, K), device="cuda")
    B = torch.randn((nb, N, K) if tb else (nb, K, N), device="cuda")
    Av = A.transpose(1, 2) if ta else A
    Bv = B.transpose(1, 2) if tb else B
    if nb == 1:
        Av, Bv = Av[0], Bv[0]
    C0 = torch.randn((nb, M, N) if nb > 1 else (M, N), device="cuda")
    for (alpha, beta) in ((1.0, 0.0), (-0.5, 0.75)):
        out = C0.clone()
        _hip.gemm(Av, Bv, out=out, alpha=alpha, beta=beta, splitk=sk)
        ref = alpha * (Av.double() @ Bv.double()) + beta * C0.double()
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        auto = lib.clo_gemm_suggest_splitk(M, N, K, nb)
        ok = err < 2e-6 * max(1, K) ** 0.5
        if not ok:
            bad += 1
        if not ok or (alpha == 1.0 and not ta and not tb):
            print(f"M={M} N={N} K={K} nb={nb} ta={ta} tb={tb} splitk={sk} (auto {auto}) alpha={alpha} beta={beta}: err {err:.2e} {'ok' if ok else 'FAIL'}")
# determinism of the stream-K fix-up
A = torch.randn(512, 2304, device="cuda"); B = torch.randn(2304, 2304, device="cuda")
r0 = _hip.gemm(A, B)
same = all(torch.equal(r0, _hip.gemm(A, B)) for _ in range(20))
print("stream-K bitwise repeatable:", same)
print("FAILURES:", bad)
