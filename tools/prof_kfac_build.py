"""One WARM ResNet-18 KFAC factor build (C4: 512 rows, joint W+b, 1 MC sample) bracketed by marker launches, for
rocprofv3 --kernel-trace (tools/run_prof_kfac_build.sh turns the trace into profiles/r03_kfac_resnet18_*)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import _hip
from benchmarks.models import ResNet18, kfac_params

if os.environ.get("KFAC_EAGER"):
    from curvlinops_amd import computers
    computers._CAPTURE = False
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
B = int(os.environ.get("ROWS", "512"))
X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
kw = dict(fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=B)
for _ in range(4):  # MIOpen find / allocator warm-up
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
torch.cuda.synchronize()
mark = torch.zeros(4099, device=dev)
_hip.axpby(mark, mark, 1.0, 0.0)   # marker: clo axpby kernel over 4099 elements
torch.cuda.synchronize()
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
torch.cuda.synchronize()
_hip.axpby(mark, mark, 1.0, 0.0)
torch.cuda.synchronize()
if os.environ.get("WITH_INVERSE"):
    Kinv = K.inverse(damping=1e-3)
    torch.cuda.synchronize()
    _hip.axpby(mark, mark, 1.0, 0.0)
    torch.cuda.synchronize()
