"""Accuracy of clo_syrk_accum_f32 (whatever kernel the dispatch picks) against float64 at the gradient-covariance
shapes of ResNet-18, for Gaussian data and for heavy-tailed rows (per-row log-normal scales, like output gradients)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
dev = torch.device("cuda:0")
lib = _hip.load()
torch.manual_seed(0)
for d, rows in ((256, 2048), (256, 1024), (256, 4096), (128, 8192), (64, 32768), (64, 131072), (512, 512), (128, 4096)):
    for kind in ("gauss", "heavy"):
        X = torch.randn(rows, d, device=dev)
        if kind == "heavy":
            X = X * torch.exp(3.0 * torch.randn(rows, 1, device=dev))
        ref = (X.double().T @ X.double())
        out = []
        for beta in (0.0,):
            C = torch.zeros(d, d, device=dev)
            _hip.syrk_accum(C, X, alpha=1.0, beta=beta)
            e = float((C.double() - ref).abs().max() / ref.abs().max())
            t = (X.T @ X)
            et = float((t.double() - ref).abs().max() / ref.abs().max())
        tall = bool(lib.clo_gram_tall_supported(rows, d, 0))
        sk = lib.clo_syrk_suggest_splitk(d, rows)
        print(f"d={d:4d} rows={rows:7d} {kind:5s}: clo {e:.1e}  (torch X^T X {et:.1e})  path: {'gram_tall' if tall else f'gemm splitk {sk}'}", flush=True)
