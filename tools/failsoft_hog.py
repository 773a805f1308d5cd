"""Keeps ~200 compute units of the GPU busy for a few seconds from ITS OWN process (one 100 KB-LDS workgroup per CU),
so that a persistent grid of another process cannot become co-resident: tests/test_gpu_kernels.py (fail-soft test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
lib = _hip.load()
torch.zeros(1, device="cuda")
st = torch.cuda.current_stream().cuda_stream
assert lib.clo_test_occupy(200, 100 * 1024, int(seconds * 1e8), st) == 0
print("hog running", flush=True)
torch.cuda.synchronize()
print("hog done", flush=True)
