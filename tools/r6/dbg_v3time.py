import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from curvlinops_amd import _hip
lib = _hip.load()
stamps = torch.zeros(1024, 8, dtype=torch.int64, device="cuda")
lib.clo_v3_timing_set.argtypes = [ctypes.c_void_p]; lib.clo_v3_timing_set.restype = None
lib.clo_v3_timing_set(ctypes.c_void_p(stamps.data_ptr()))
A = torch.randn(1024, 1024, device="cuda"); B = torch.randn(1024, 1024, device="cuda"); out = torch.empty(1024, 1024, device="cuda")
_hip.gemm(A, B, out=out); torch.cuda.synchronize()
print("nonzero", int((stamps != 0).sum()), stamps[:2].cpu())
