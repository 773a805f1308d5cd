"""Phase timeline of mid_fused_fwd_kernel (-DCLO_MFU_TIMING build): wall_clock64 stamps (100 MHz) of every wave of workgroups 0, 64, 128, 192."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from curvlinops_amd import _hip
lib = _hip.load()
dims, acts = [1024, 2688, 2688, 10], [1, 1, 0]
torch.manual_seed(0)
W = [torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5 for i in range(3)]
b = [torch.randn(dims[i + 1], device="cuda") * 0.1 for i in range(3)]
VW = [torch.rand_like(w) for w in W]; Vb = [torch.rand_like(x) for x in b]
OW = [torch.empty_like(w) for w in W]; Ob = [torch.empty_like(x) for x in b]
plan = _hip.MLPPlan(dims, acts)
for N in [int(a) for a in sys.argv[1:]] or [16, 64]:
    X = torch.rand(N, dims[0], device="cuda")
    for i in range(6):
        plan.ggn_matvec(W, b, VW, Vb, OW, Ob, X, 0, 2.0 / (N * 10), 1.0, 0.0)
    torch.cuda.synchronize()
    ws = plan.workspace(N, X.device)
    tail = ws[-4096:].view(torch.int64).cpu().numpy().reshape(-1, 16)   # [(block/64) * 8 + wave][16]
    t0 = min(int(r[0]) for r in tail[:32] if r[0] > 0)
    names = ["start", "l1 mfma done", "merged", "xw stores issued", "xw drained", "xw seam done", "after seam barrier", "staged issued+written", "stage barrier", "chunk1 done", "chunk2 done", "end"]
    print(f"N={N}: stamps in us from the first start; rows = (workgroup, wave)")
    for k in range(32):
        r = tail[k]
        print(f"  wg {64 * (k // 8):3d} wave {k % 8}: " + " ".join(f"{names[i][:6]}={(int(r[i]) - t0) / 100.0:5.1f}" if r[i] > 0 else f"{names[i][:6]}=  -  " for i in range(12)))
