"""A/B of the <= 8-row chains on C2 through the public operator API (env switches are read per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
params = dict(model.named_parameters())
rows = int(os.environ.get("ROWS", "8"))
X, y = torch.rand(rows, 1024, device=dev), torch.rand(rows, 10, device=dev)
G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
D = G.shape[1]
vs = [torch.rand(D, device=dev) for _ in range(8)]


def run(n):
    for i in range(10):
        G @ vs[i % 8]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = G @ vs[i % 8]
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / n, 1e6 * (time.perf_counter() - t0) / n


# FLAGS=<int> variants set the operator's CLO_MLP_* kernel choice (1 = launch chain), anything else goes to the environment
variants = [v.split(",") for v in os.environ.get("VARIANTS", "FLAGS=1;FLAGS=0").split(";")]
for rnd in range(3):
    for var in variants:
        for kv in var:
            k, v = kv.split("=")
            if k == "FLAGS":
                G.native_flags = int(v)
            else:
                os.environ[k] = v
        host, tot = run(300)
        print(f"round {rnd} {' '.join(var):40s} host {host:6.1f} us  total {tot:6.1f} us/matvec", flush=True)
ref = None
for var in variants:
    for kv in var:
        k, v = kv.split("=")
        if k == "FLAGS":
            G.native_flags = int(v)
        else:
            os.environ[k] = v
    out = G @ vs[0]
    if ref is None:
        ref = out
    print(" ".join(var), "rel diff vs first variant", float((out - ref).abs().max() / ref.abs().max()))
