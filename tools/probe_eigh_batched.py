"""Scratch: torch.linalg.eigh (rocSOLVER) on a stacked batch of equal-size factors vs one by one."""
import sys, os, time
import torch
dev = torch.device("cuda:0")
def t(fn, reps=2):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for n, b in ((64, 8), (512, 5), (577, 5), (1153, 4), (2305, 4), (4609, 3), (768, 12), (3072, 6)):
    mats = []
    for _ in range(b):
        X = torch.randn(n + 8, n, device=dev); mats.append(X.T @ X / n)
    S = torch.stack(mats)
    t_seq = t(lambda: [torch.linalg.eigh(M) for M in mats])
    t_bat = t(lambda: torch.linalg.eigh(S))
    print(f"n={n:5d} batch={b:2d}: one by one {t_seq:8.1f} ms | stacked {t_bat:8.1f} ms")
