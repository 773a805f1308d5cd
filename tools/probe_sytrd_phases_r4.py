"""Round 4: per-column phase times of the persistent panel kernel (library built with -DCLO_TD_TIMING):
CLO_HIP_LIB=curvlinops_amd/lib/variants/libclo_tdt.so python tools/probe_sytrd_phases_r4.py [n] [max_blocks]
(stamps of the first and the last workgroup, summed over the 64 columns of the FIRST panel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from curvlinops_amd import _hip
lib = _hip.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4609
maxb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
names = ["exchange loads", "reflector", "careful path", "row pass", "partial sums", "store ack", "barrier"]
for kind in ("lowrank", "wishart"):
    r = max(16, n // 3) if kind == "lowrank" else 2 * n
    X = torch.rand(r, n, generator=g).to(dev); A = X.T @ X / r; A = A / A.abs().max()
    ld = (n + 3) // 4 * 4
    n4 = ld
    work = torch.zeros(n, ld, device=dev); work[:, :n] = A
    D, E, tau = (torch.zeros(n, device=dev) for _ in range(3))
    nb = lib.clo_sytrd_ws_bytes(n); ws = torch.zeros(nb // 4, device=dev)
    rc = lib.clo_sytrd_f32(work.data_ptr(), ld, n, D.data_ptr(), E.data_ptr(), tau.data_ptr(), ws.data_ptr(), nb, maxb,
                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); assert rc == 0
    off = 2 * n * 64 + 4 * n4 + 192 + 3 * 16 * 264 + 1088   # (+ TP_CNT_WORDS of sytrd.hip)
    st = ws[off:off + 32].view(torch.int64).cpu().numpy().astype(np.float64) * 0.01 / 64   # us per column
    print(f"n={n} {kind}: us per column, first / last workgroup")
    for i, nm in enumerate(names):
        print(f"   {nm:16s} {st[i]:7.2f} {st[8 + i]:7.2f}")
    print(f"   {'total':16s} {st[:7].sum():7.2f} {st[8:15].sum():7.2f}")
    pw = ws[off + 32:off + 32 + 512].view(torch.int64).cpu().numpy()
    rp, br = (pw >> 32).astype(np.float64) * 0.01 / 64, (pw & 0xffffffff).astype(np.float64) * 0.01 / 64
    print("   row pass per workgroup (us per column), workgroups 0, 8, 16, ...:", " ".join(f"{x:.1f}" for x in rp[::8]))
    print("   row pass by XCD (workgroup % 8), mean:", " ".join(f"{rp[x::8][rp[x::8] > 0].mean():.1f}" for x in range(8)))
    print("   barrier wait by workgroup 0, 8, ...:", " ".join(f"{x:.1f}" for x in br[::8]))
