"""Scratch: time single fwd layer launches via the C ABI with preallocated buffers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
lib = _hip.load()
st = torch.cuda.current_stream().cuda_stream
for (di, do, da) in [(2688, 2688, True), (1024, 2688, False), (1040, 2688, False), (1088, 2688, False), (2048, 2688, False), (2064, 2688, False), (2048, 2688, True), (2064, 2688, True)]:
    nb = 6
    W = [torch.randn(do, di, device="cuda") for _ in range(nb)]
    V = [torch.randn(do, di, device="cuda") for _ in range(nb)]
    a = torch.rand(8, di, device="cuda"); d = torch.rand(8, di, device="cuda") if da else None
    b = torch.rand(do, device="cuda")
    ao, dao, dph = (torch.empty(8, do, device="cuda") for _ in range(3))
    ws = torch.empty(lib.clo_mlp_fwd_ws_floats(8, di, do), device="cuda")
    def call(i):
        rc = lib.clo_mlp_fwd_jvp_layer(W[i%nb].data_ptr(), b.data_ptr(), V[i%nb].data_ptr(), b.data_ptr(), a.data_ptr(),
             d.data_ptr() if d is not None else None, ao.data_ptr(), dao.data_ptr(), dph.data_ptr(), 8, di, do, 1, ws.data_ptr(), st)
        assert rc == 0
    for i in range(3): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 60
    e0.record()
    for i in range(n): call(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{os.environ.get('CLO_HIP_LIB','default')[-14:]:>14} fwd {di}->{do}: {us:.1f} us  {8*di*do/us/1e6:.2f} TB/s")
