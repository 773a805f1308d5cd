"""Stress of the pipelined Cholesky inverse (orders >= 1536): random sizes (block-aligned and not), single / batched /
concurrent (operator-level batch with worker threads), against a float64 inverse on the device, and run-to-run bit equality
(a race between the streams of the pipeline would show up as differing results).
    python tools/stress_chol_pipeline.py [seed] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip, linalg_native as L
_hip.load()
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = torch.Generator(device="cpu").manual_seed(seed)
sizes = [1536, 1537, 1600, 1664, 1789, 2048, 2305, 2500, 3072, 3333, 4100, 4609]

def spd(n):
    X = torch.randn(n + 64, n, generator=g).to(dev)
    return X.T @ X / (n + 64)

def check(A, X, damping, what):
    A64 = A.double() + damping * torch.eye(A.shape[0], device=dev, dtype=torch.float64)
    # (residual instead of a float64 inverse: hipBLAS' double-precision trsm ran out of workspace next to our buffers)
    R = A64 @ X.double()
    R.diagonal().sub_(1.0)
    err = float(R.abs().max())
    assert err < 2e-3 and torch.isfinite(X).all(), f"{what}: n={A.shape[0]} residual {err:.2e}"
    return err

worst = 0.0
for r in range(rounds):
    # single calls, each twice: bit-identical
    for n in [sizes[int(i)] for i in torch.randint(0, len(sizes), (3,), generator=g)]:
        A = spd(n)
        X1 = _hip.cholesky_inverse(A, 1e-3)
        X2 = _hip.cholesky_inverse(A, 1e-3)
        assert torch.equal(X1, X2), f"single n={n}: run-to-run difference {float((X1 - X2).abs().max()):.2e}"
        worst = max(worst, check(A, X1, 1e-3, "single"))
    # batched equal sizes
    n = sizes[int(torch.randint(0, 8, (1,), generator=g))]
    mats = [spd(n) for _ in range(3)]
    outs = [torch.empty_like(m) for m in mats]
    outs2 = [torch.empty_like(m) for m in mats]
    st = torch.zeros(3, device=dev, dtype=torch.int32)
    _hip.cholesky_inverse_batched_into(mats, [1e-3, 2e-3, 5e-4], outs, st)
    _hip.cholesky_inverse_batched_into(mats, [1e-3, 2e-3, 5e-4], outs2, st)
    for A, X, X2, d in zip(mats, outs, outs2, (1e-3, 2e-3, 5e-4)):
        assert torch.equal(X, X2), f"batched n={n}: run-to-run difference"
        worst = max(worst, check(A, X, d, "batched"))
    # operator-level batch: worker threads, mixed sizes (pipelined and single-chain units side by side)
    mix = [spd(s) for s in (2305, 2305, 1600, 1153, 577, 577, 3072, 64, 129)]
    with L.concurrent_inverses():
        res = [L.damped_cholesky_inverse(A, 1e-3) for A in mix]
    with L.concurrent_inverses():
        res2 = [L.damped_cholesky_inverse(A, 1e-3) for A in mix]
    torch.cuda.synchronize()
    for A, X, X2 in zip(mix, res, res2):
        assert torch.equal(X, X2), f"concurrent n={A.shape[0]}: run-to-run difference {float((X - X2).abs().max()):.2e}"
        worst = max(worst, check(A, X, 1e-3, "concurrent"))
    print(f"round {r}: ok (worst |A X - I| so far {worst:.2e})", flush=True)
print("done")
