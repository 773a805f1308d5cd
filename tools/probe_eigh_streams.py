"""ResNet-18 (C4, 512 rows) EKFAC bases, native solver: eigh_many time against the number of worker streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import linalg_native as L
from benchmarks.models import ResNet18, kfac_params

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc",
                         separate_weight_and_bias=False, check_deterministic=False, num_data=512)
facs = [S for blk in K[1] for S in blk]
print("sizes:", sorted(int(S.shape[0]) for S in facs))
for ns in [int(a) for a in sys.argv[1:]] or (4, 5, 6, 7, 8, 10, 6, 8):
    L.eigh_many(facs, num_streams=ns); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); out = L.eigh_many(facs, num_streams=ns); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"streams {ns:2d}: eigh_many {1e3*best:7.1f} ms")
