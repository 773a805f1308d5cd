"""Phase timeline of the persistent <= 8-row kernel (library built with -DCLO_MEGA_TIMING, see tools/buildvar.sh):
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_timing.so python tools/probe_mega_timing.py"""
import os, sys
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
plan = _hip.MLPPlan(dims, acts)
N = 8
X = torch.rand(N, dims[0], device="cuda")
names = ["entry", "loads issued", "L1 mfma+merge", "L1 epilogue", "seamA a1 gathered", "L2 mfma", "slab pub+rowA",
         "finish+hp pub", "top wait", "loss", "delta2", "delta1 mfma", "slab2 pub", "write-only", "colB wait", "end",
         "w7 merge start", "w7 merge done", "w7 publish issued", "w7 gather complete", "w7 a1 in LDS", "sweep done (wave 0)", "sweep barrier"] + ["-"] * 9
acc = []
B2B = int(os.environ.get("BACK2BACK", "1"))   # > 1: that many products queued back to back, the last one's stamps are read
for i in range(30):
    for j in range(B2B):
        k = (i * B2B + j) % nv
        plan.ggn_matvec(W, b, VW[k], Vb[k], OW[k], Ob[k], X, 0, 2.0 / 80, 1.0, 0.0)
    torch.cuda.synchronize()
    ws = next(iter(plan._ws.values()))
    t = ws[-16384:].view(torch.int64).cpu().numpy()[:256 * 32].reshape(256, 32).astype(np.float64) * 0.01  # 100 MHz -> us
    if i >= 10:
        acc.append(t - t[:, :1].min())
t = np.mean(acc, axis=0)
print("stamp                  min     mean      max   (us since the first workgroup's entry; mean over 20 calls)")
for i, n in enumerate(names):
    if n == "-" or t[:, i].max() <= 0:
        continue
    print(f"{i:2d} {n:18s} {t[:, i].min():7.2f}  {t[:, i].mean():7.2f}  {t[:, i].max():7.2f}")

print("entry by XCD (workgroup w runs on XCD w % 8):", " ".join(f"{t[x::8, 0].mean():.2f}" for x in range(8)))
print("end   by XCD:", " ".join(f"{t[x::8, 15].mean():.2f}" for x in range(8)))
out = os.environ.get("STAMPS_OUT")
if out:
    np.save(out, t)   # [256 workgroups][32 stamps], mean over the measured calls
