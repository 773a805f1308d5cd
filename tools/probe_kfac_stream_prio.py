"""KFAC factor build (ResNet-18, 512 rows): does the priority of the factor stream (im2col + SYRK beside autograd) matter?
GPU_MAX_HW_QUEUES / FACTOR_PRIO from the environment; prints the min / median of 9 builds after 3 warm-ups."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import computers
from benchmarks.models import ResNet18, kfac_params

dev = torch.device("cuda:0")
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
prio = os.environ.get("FACTOR_PRIO")
if prio is not None:
    computers._FACTOR_STREAMS[dev] = torch.cuda.Stream(device=dev, priority=int(prio))
torch.manual_seed(0)
model = ResNet18(num_classes=10).to(dev).eval()
params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
def build():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", mc_samples=1,
                             separate_weight_and_bias=False, check_deterministic=False)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
for _ in range(3): build()
ts = sorted(build() for _ in range(9))
print(f"queues {os.environ.get('GPU_MAX_HW_QUEUES', '4')} factor-stream priority {prio}: build min {ts[0]:.2f} median {ts[4]:.2f} max {ts[-1]:.2f} ms")
