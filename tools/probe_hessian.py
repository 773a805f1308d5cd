import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
params = dict(model.named_parameters())
for N in (8, 128, 512):
    X, y = torch.rand(N, 1024, device=dev), torch.rand(N, 10, device=dev)
    for native in (True, False):
        H = C.HessianLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
        if not native: H._native = None
        v = torch.rand(H.shape[1], device=dev)
        for _ in range(3): H @ v
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): H @ v
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
        print(f"C2 Hessian matvec N={N} {'native' if native else 'autograd'}: {t*1e6:.0f} us")
