"""Scratch: one GEMM shape repeated (for PMC passes):  python tools/probe_gemm_one.py M N K [splitk]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
M, N, K = (int(x) for x in sys.argv[1:4]); s = int(sys.argv[4]) if len(sys.argv) > 4 else None
A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
for _ in range(10): _hip.gemm(A, B, out=out, splitk=s)
torch.cuda.synchronize()
