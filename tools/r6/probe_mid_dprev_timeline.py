"""Per-block timeline of mid_dprev_kernel (delta_{l-1} = delta_l W_l of the 9 ... 64-row chain; -DCLO_MID_TIMING build)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from curvlinops_amd import _hip
lib = _hip.load()
stamps = torch.zeros(8192, 8, dtype=torch.int64, device="cuda")
lib.clo_mid_timing_set.argtypes = [ctypes.c_void_p]; lib.clo_mid_timing_set.restype = None
lib.clo_mid_timing_set(ctypes.c_void_p(stamps.data_ptr()))
dims, acts = [1024, 2816, 2688, 10], [1, 1, 0]
torch.manual_seed(0)
W = [torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5 for i in range(3)]
b = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(3)]
VW = [torch.rand_like(w) for w in W]; Vb = [torch.rand_like(x) for x in b]
OW = [torch.empty_like(w) for w in W]; Ob = [torch.empty_like(x) for x in b]
plan = _hip.MLPPlan(dims, acts)
for N in [int(a) for a in sys.argv[1:]] or [16, 32, 48, 64]:
    X = torch.rand(N, dims[0], device="cuda")
    for i in range(6):
        plan.ggn_matvec(W, b, VW, Vb, OW, Ob, X, 0, 2.0 / (N * 10), 1.0, 0.0)
    torch.cuda.synchronize()
    stamps.zero_(); torch.cuda.synchronize()
    plan.ggn_matvec(W, b, VW, Vb, OW, Ob, X, 0, 2.0 / (N * 10), 1.0, 0.0)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()[6144:]
    live = s[:, 0] > 0
    nb = int(live.sum())
    t0 = s[live, 0].min()
    ent, stg, lp, mg = [(s[live, i] - t0) / 100.0 for i in range(4)]
    print(f"N={N}: {nb} blocks; last merge done at {mg.max():.2f} us")
    print(f"   entry      min/mean/max {ent.min():6.2f} {ent.mean():6.2f} {ent.max():6.2f}")
    print(f"   staging    mean {np.mean(stg - ent):5.2f} max {np.max(stg - ent):5.2f} us   (entry -> delta in LDS; the first weight loads are in flight)")
    print(f"   row loop   mean {np.mean(lp - stg):5.2f} min {np.min(lp - stg):5.2f} max {np.max(lp - stg):5.2f} us   (wave 0 of the block)")
    print(f"   merge      mean {np.mean(mg - lp):5.2f} max {np.max(mg - lp):5.2f} us")
