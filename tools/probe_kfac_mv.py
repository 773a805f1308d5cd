import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18().to(dev).eval(); params = kfac_params(model); B = 512
X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False,
                         check_deterministic=False, num_data=B)
for k in (1, 8, 32):
    v = torch.rand(K.shape[1], k, device=dev) if k > 1 else torch.rand(K.shape[1], device=dev)
    for _ in range(3): K @ v
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): r = K @ v
    t_host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"K={k}: {t*1e3:.2f} ms per product (host enqueue {t_host*1e3:.2f} ms)")
import cProfile, pstats
v = torch.rand(K.shape[1], device=dev)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): K @ v
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
