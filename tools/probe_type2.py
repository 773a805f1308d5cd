import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
import curvlinops_amd.computers as CC
from benchmarks.models import lenet5, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = lenet5().to(dev).eval(); params = kfac_params(model); B = 1024
X, y = torch.rand(B, 1, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
        T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
orig_grad = torch.autograd.grad
def timed_grad(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        r = orig_grad(*a, **k)
    except Exception as e:
        print("autograd.grad raised:", type(e).__name__, str(e)[:200]); raise
    torch.cuda.synchronize(); T["autograd.grad"] = T.get("autograd.grad", 0) + time.perf_counter() - t0; return r
torch.autograd.grad = timed_grad
for i in range(3):
    T.clear()
    comp = CC.HipKFACComputer(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="type-2", separate_weight_and_bias=False, check_deterministic=False, num_data=B)
    wrap(comp, "_grad_outputs_computer")
    torch.cuda.synchronize(); t0 = time.perf_counter(); comp.compute(); torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"total {tot*1e3:.1f} ms", {k: round(v*1e3, 2) for k, v in T.items()})
