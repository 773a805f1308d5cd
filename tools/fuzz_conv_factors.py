"""Randomised check of the Conv2d input-covariance kernels: random geometry (kernel, stride, padding, dilation,
with / without the bias column), both routes -- patches generated inside the symmetric GEMM's tile loader
(clo_im2col_syrk_accum_f32) and materialised patches (clo_im2col_f32 + clo_syrk_accum_f32 / the tall-skinny Gram
kernel) -- against F.unfold in float64.    python tools/fuzz_conv_factors.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from curvlinops_amd import _hip


def run(seed, ncase):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst, fails = 0.0, []
    for case in range(ncase):
        B, C = int(rng.integers(1, 9)), int(rng.integers(1, 20))
        H, W = int(rng.integers(3, 30)), int(rng.integers(3, 30))
        ks = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
        dl = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        st = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        pd = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        OH = (H + 2 * pd[0] - dl[0] * (ks[0] - 1) - 1) // st[0] + 1
        OW = (W + 2 * pd[1] - dl[1] * (ks[1] - 1) - 1) // st[1] + 1
        if OH < 1 or OW < 1:
            continue
        ones = bool(rng.random() < 0.5)
        beta = float(rng.choice([0.0, 1.0, 0.5]))
        alpha = float(rng.uniform(0.1, 2.0))
        x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(case + 1000 * seed))
        P = F.unfold(x.double(), ks, dilation=dl, padding=pd, stride=st).transpose(1, 2).reshape(B * OH * OW, -1)
        if ones:
            P = torch.cat([P, torch.ones(P.shape[0], 1, dtype=torch.float64)], dim=1)
        d = P.shape[1]
        C0 = torch.randn(d, d, generator=torch.Generator().manual_seed(case))
        C0 = C0 + C0.T
        ref = beta * C0.double() + alpha * (P.T @ P)
        xd = x.to(dev)
        what = f"case {case}: B={B} C={C} HxW={H}x{W} k={ks} s={st} p={pd} d={dl} ones={ones} beta={beta}"
        # fused
        Cf = C0.to(dev).clone()
        _hip.im2col_syrk_accum(Cf, xd, ks, st, pd, dl, alpha=alpha, beta=beta, ones_col=ones)
        # materialised
        Cm = C0.to(dev).clone()
        pat = _hip.im2col(xd, ks, st, pd, dl).reshape(B * OH * OW, -1)
        _hip.syrk_accum(Cm, pat, alpha=alpha, beta=beta, ones_col=ones)
        sc = float(ref.abs().max())
        for name, got in (("fused", Cf), ("materialised", Cm)):
            err = float((got.double().cpu() - ref).abs().max()) / sc
            worst = max(worst, err)
            if not err < 2e-5:
                fails.append(f"{what}: {name} err {err:.1e}")
    return worst, fails


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    worst, fails = run(seed, ncase)
    for f in fails:
        print(f)
    print(f"done: {ncase} cases, worst rel err {worst:.2e}, {len(fails)} failures")
