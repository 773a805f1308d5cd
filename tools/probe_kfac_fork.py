"""Captured ResNet-18 factor build: one fork of the factor stream per hook ("fine") against one fork per batch ("coarse")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import computers
from benchmarks.models import ResNet18, kfac_params

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
g = torch.Generator().manual_seed(4321)
X = torch.rand(512, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (512,), generator=g).to(dev)
kw = dict(fisher_type="empirical", separate_weight_and_bias=False, check_deterministic=False, num_data=512)
def build():
    return C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
computers._CAPTURE = False
ref = [f.clone() for blk in build()[1] for f in blk]
computers._CAPTURE = True
for mode in ("coarse_inline", "coarse_tail", "coarse_chunk6", "fine", "serial"):
    computers._CAPTURE_FORK = "fine" if mode in ("serial", "fine") else "coarse"
    computers._CAPTURE_G_CHUNK = {"coarse_inline": 0, "coarse_tail": 10**6, "coarse_chunk6": 6}.get(mode, 0)
    computers._OVERLAP = mode != "serial"
    computers._CAPTURE_BRANCHES = 1 if mode == "serial" else 2   # (round 6: the package's own rule is by queue count; here every form is forced)
    computers.reset_captured_builds()
    ts = []
    for i in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); K = build(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    fac = [f for blk in K[1] for f in blk]
    worst = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(ref, fac))
    print(f"queues={sys.argv[1]} fork={mode}: " + " ".join(f"{t:.2f}" for t in ts[2:]) + f" ms | vs eager {worst:.1e}", flush=True)
