"""Kronecker products with factors in row- and column-major storage (torch.linalg.eigh returns column-major
eigenvectors), odd sizes: _kron_apply_native vs dense float64."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd.kronecker import _kron_apply_native
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (a, b, K) in ((4, 845, 3), (4, 845, 1), (5, 27, 3), (3, 4, 3), (845, 4, 3), (64, 128, 4), (7, 577, 2)):
    for lay1, lay2, tr in itertools.product(("row", "col"), ("row", "col"), (False, True)):
        S1 = torch.randn(a, a, device=dev); S2 = torch.randn(b, b, device=dev)
        if lay1 == "col": S1 = S1.T.contiguous().T
        if lay2 == "col": S2 = S2.T.contiguous().T
        x = torch.randn(a * b, K, device=dev)
        got = _kron_apply_native([S1, S2], x, tr)
        M = torch.kron(S1.double().cpu().contiguous(), S2.double().cpu().contiguous())
        ref = (M.T if tr else M) @ x.double().cpu()
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        flag = "" if err < 1e-4 else "   <-- WRONG"
        print(f"a={a} b={b} K={K} S1 {lay1} S2 {lay2} transpose={tr}: err {err:.1e}{flag}")
