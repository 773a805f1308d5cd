"""Why do the damped inverses take 17 ms in bench.py's kfac leg and 11.7 ms in tools/probe_queues.py?  Same process setup as
bench.py; variants chosen by argv[1]: plain | fwdbwd (run the plain gradient passes first) | nocapture."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("HWQ", "16")
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import computers
from benchmarks.models import ResNet18, kfac_params

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
g = torch.Generator(device="cpu").manual_seed(4321)
X = torch.rand(512, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (512,), generator=g).to(dev)
kw = dict(fisher_type="mc", mc_samples=1, separate_weight_and_bias=False, check_deterministic=False, num_data=512)
if mode == "nocapture":
    computers._CAPTURE = False
if mode == "serialcap":
    computers._OVERLAP = False
if mode == "poolstream":
    computers._CAPTURE_STREAM = 1
if mode == "coarsetail":
    computers._CAPTURE_G_CHUNK = 10**6
if mode == "finecap":
    computers._CAPTURE_FORK = "fine"
if mode == "warmchol":   # create the Cholesky pipeline's helper streams BEFORE any graph is captured
    from curvlinops_amd import _hip
    A0 = torch.rand(2048, 2048, device=dev); A0 = A0 @ A0.T + 2048 * torch.eye(2048, device=dev)
    _hip.cholesky_inverse(A0, 0.0); torch.cuda.synchronize()
for _ in range(6):
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
torch.cuda.synchronize()
if mode == "fwdbwd":
    for _ in range(6):
        o = model(X)
        torch.autograd.grad(nn.functional.cross_entropy(o, y), list(params.values()))
    torch.cuda.synchronize()
if mode == "resetonly":
    import gc
    computers.reset_captured_builds(); gc.collect(); torch.cuda.synchronize()
if mode == "emptyonly":
    torch.cuda.empty_cache(); torch.cuda.synchronize()
if mode == "resetcap":   # captured builds, but the graphs (their memory pools, their streams) are gone before the inverses
    import gc
    computers.reset_captured_builds(); gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
if mode == "clonefac":   # captured builds, but the factors are re-homed into separately allocated tensors
    for blk in K[1]:
        for i in range(len(blk._factors) if hasattr(blk, "_factors") else 0):
            blk._factors[i] = blk._factors[i].clone()
    print("factor attr:", [a for a in dir(K[1][0]) if "factor" in a.lower()][:6], flush=True)
v = torch.rand(K.shape[1], device=dev)
for _ in range(6):
    K @ v
torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); Kinv = K.inverse(damping=1e-3); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
if os.environ.get("MARK"):
    from curvlinops_amd import _hip
    mark = torch.zeros(4099, device=dev)
    _hip.axpby(mark, mark, 1.0, 0.0); torch.cuda.synchronize()
    Kinv = K.inverse(damping=1e-3); torch.cuda.synchronize()
    _hip.axpby(mark, mark, 1.0, 0.0); torch.cuda.synchronize()
print(f"{mode}: inverse calls " + " ".join(f"{t:.1f}" for t in ts) + " ms", flush=True)
