"""Scratch: synchronised per-call latency of the (batched) Cholesky inverse, pipeline on/off (CLO_CHOL_PIPE)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
dev = torch.device("cuda:0")
for n, batch in ((4609, 1), (4609, 3), (2305, 1), (2305, 4), (1153, 4)):
    mats = []
    for b in range(batch):
        X = torch.randn(2 * n, n, device=dev)
        mats.append(X.T @ X / (2 * n))
    outs = [torch.empty_like(m) for m in mats]
    status = torch.zeros(batch, device=dev, dtype=torch.int32)
    def run():
        if batch == 1:
            return _hip.cholesky_inverse_async(mats[0], 1e-3)
        _hip.cholesky_inverse_batched_into(mats, [1e-3] * batch, outs, status)
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"PIPE={os.environ.get('CLO_CHOL_PIPE', '1')} n={n} batch={batch}: " + " ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)
