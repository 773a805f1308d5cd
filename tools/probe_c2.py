"""Scratch probe: time the native C2 GGN matvec (kernel path only) at several batch sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from curvlinops_amd import _hip

dims, acts = [1024, 2688, 2688, 10], [1, 1, 0]
torch.manual_seed(0)
W = [torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5 for i in range(3)]
b = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(3)]
nv = 8
VW = [[torch.rand_like(w) for w in W] for _ in range(nv)]
Vb = [[torch.rand_like(x) for x in b] for _ in range(nv)]
OW = [[torch.empty_like(w) for w in W] for _ in range(nv)]
Ob = [[torch.empty_like(x) for x in b] for _ in range(nv)]
D = sum(w.numel() for w in W) + sum(x.numel() for x in b)
plan = _hip.MLPPlan(dims, acts)
for N in [int(a) for a in sys.argv[1:]] or [8, 16, 128, 512]:
    X = torch.rand(N, dims[0], device="cuda")
    scale = 2.0 / (N * dims[-1])
    def step(i):
        k = i % nv
        plan.ggn_matvec(W, b, VW[k], Vb[k], OW[k], Ob[k], X, 0, scale, 1.0, 0.0)
    for i in range(5): step(i)
    torch.cuda.synchronize()
    steps = 200 if N <= 16 else 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(steps): step(i)
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / steps
    print(f"N={N}: {ms*1e3:.1f} us/matvec (host {1e6*(t1-t0)/steps:.1f} us)  "
          f"alg GB/s={12*D/ms/1e6:.0f}  alg TFLOP/s={10*N*D/ms/1e9:.1f}")
