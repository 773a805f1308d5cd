"""rows128 / rows512 of the bench line (operator API, G @ v) against the kernel path alone (MLPPlan.ggn_matvec), same process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import curvlinops_amd as C
import bench
device = torch.device("cuda")
for rows in (128, 512):
    model, X, y = bench.build_problem(device, rows, seed=0)
    params = dict(model.named_parameters())
    D = sum(p.numel() for p in params.values())
    G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
    v = torch.rand(D, device=device)
    for _ in range(3): G @ v
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(30): G @ v
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"rows={rows}: G @ v {1e3 * e0.elapsed_time(e1) / 30:.1f} us (events) {1e6 * (t1 - t0) / 30:.1f} us (host)")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): G @ v
        torch.cuda.synchronize()
    rowsk = sorted(prof.key_averages(), key=lambda a: -a.device_time_total)[:14]
    for a in rowsk: print(f"   {a.count:4d} {a.device_time_total / 5:9.1f} us/matvec  {a.key[:90]}")
