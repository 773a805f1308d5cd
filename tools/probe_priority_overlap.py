"""Does a high-priority HIP stream let a compute-bound GEMM run at full speed beside an HBM-bound stream kernel?
(round 4, K-column products: the delta GEMMs beside the result streams).  Stand-ins: a 1 GiB torch copy kernel on the
default-priority stream, clo GEMM 2688 x 2688 x 256 on a second stream with priority 0 / -1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip

dev = torch.device("cuda:0")
W = torch.randn(2688, 2688, device=dev)
B = torch.randn(2688, 256, device=dev)
out = torch.empty(2688, 256, device=dev)
big = torch.empty(1 << 28, device=dev)       # 1 GiB
src = torch.randn(2688, 8, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


def stream_kernel():
    big.fill_(1.5)        # write stream: 1 GiB


def gemms(n=4):
    for _ in range(n):
        _hip.gemm(W, B, out=out)


print(f"write stream alone {timed(stream_kernel):8.1f} us   4 GEMMs alone {timed(gemms):8.1f} us")
main = torch.cuda.current_stream()
for prio in (0, -1):
    side = torch.cuda.Stream(priority=prio)
    ev_g = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def both():
        side.wait_stream(main)
        stream_kernel()
        with torch.cuda.stream(side):
            ev_g[0].record()
            gemms()
            ev_g[1].record()
        main.wait_stream(side)

    t = timed(both)
    torch.cuda.synchronize()
    print(f"priority {prio:2d}: both {t:8.1f} us   (4 GEMMs inside: {1e3 * ev_g[0].elapsed_time(ev_g[1]):8.1f} us)")
