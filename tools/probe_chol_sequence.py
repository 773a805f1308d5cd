"""Scratch: the batched Cholesky inverses of ResNet-18's big factor groups back to back on one stream (no sync between)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
dev = torch.device("cuda:0")
groups = []
for n, batch in ((4609, 3), (2305, 4), (1153, 4), (577, 5)):
    mats = []
    for b in range(batch):
        X = torch.randn(2 * n, n, device=dev)
        mats.append(X.T @ X / (2 * n))
    groups.append((mats, [torch.empty_like(m) for m in mats], torch.zeros(batch, device=dev, dtype=torch.int32)))
def run(sync):
    for mats, outs, status in groups:
        _hip.cholesky_inverse_batched_into(mats, [1e-3] * len(mats), outs, status)
        if sync: torch.cuda.synchronize()
for sync in (False, True):
    ts = []
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(sync); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"PIPE={os.environ.get('CLO_CHOL_PIPE', '1')} sync between units={sync}: " + " ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)

for it in range(3):
    parts = []
    for mats, outs, status in groups:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _hip.cholesky_inverse_batched_into(mats, [1e-3] * len(mats), outs, status)
        torch.cuda.synchronize(); parts.append((time.perf_counter() - t0) * 1e3)
    print(f"PIPE={os.environ.get('CLO_CHOL_PIPE', '1')} per unit: " + " ".join(f"{t:.2f}" for t in parts), flush=True)
