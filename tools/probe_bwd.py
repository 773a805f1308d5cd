"""Scratch: time bwd layer variants (L2 shape) via the C ABI with preallocated buffers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
lib = _hip.load()
st = torch.cuda.current_stream().cuda_stream
di, do, N = 2688, 2688, 8
nb = 6
W = [torch.randn(do, di, device="cuda") for _ in range(nb)]
O = [torch.empty(do, di, device="cuda") for _ in range(nb)]
ob = torch.empty(do, device="cuda")
delta = torch.rand(N, do, device="cuda"); a = torch.rand(N, di, device="cuda"); dphi = torch.rand(N, di, device="cuda")
dprev = torch.empty(N, di, device="cuda")
ws = torch.zeros(lib.clo_mlp_bwd_ws_floats(N, di, do), device="cuda")
def run(outer, dp, beta=0.0):
    def call(i):
        rc = lib.clo_mlp_bwd_layer(W[i%nb].data_ptr(), delta.data_ptr(), a.data_ptr(), dphi.data_ptr(),
              O[i%nb].data_ptr() if outer else None, ob.data_ptr() if outer else None, dprev.data_ptr() if dp else None,
              1.0, beta, N, di, do, ws.data_ptr(), st)
        assert rc == 0
    for i in range(3): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 60
    e0.record()
    for i in range(n): call(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"outer+dprev: {run(True, True):.1f} us   outer only: {run(True, False):.1f} us   dprev only: {run(False, True):.1f} us   outer+dprev beta=1: {run(True, True, 1.0):.1f} us")
