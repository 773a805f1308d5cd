import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params, lenet5
which = sys.argv[1]
dev = torch.device("cuda:0"); torch.manual_seed(0)
model, shape, B = (lenet5(), (1, 32, 32), 1024) if which == "lenet" else (ResNet18(), (3, 32, 32), 512)
model = model.to(dev).eval(); params = kfac_params(model)
X, y = torch.rand(B, *shape, device=dev), torch.randint(0, 10, (B,), device=dev)
for _ in range(3):
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=B)
torch.cuda.synchronize()
if len(sys.argv) > 2:
    for _ in range(2): K.inverse(damping=1e-3)
    torch.cuda.synchronize()
