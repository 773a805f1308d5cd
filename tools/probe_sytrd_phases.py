"""Wall time of clo_sytrd_f32 (set CLO_TD_DEBUG to skip kernel phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [577, 1153, 4609]:
    ld = (n + 3) // 4 * 4
    X = torch.randn(n, n, device=dev)
    A0 = torch.zeros(n, ld, device=dev); A0[:, :n] = X @ X.T / n
    best = 1e9
    for _ in range(3):
        A = A0.clone(); torch.cuda.synchronize(); t = time.perf_counter()
        _hip.sytrd_(A, n); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"dbg={os.environ.get('CLO_TD_DEBUG','0'):>3s} n={n}: {best*1e3:7.2f} ms  {best*1e6/n:6.2f} us/col", flush=True)
