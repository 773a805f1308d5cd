"""Host-side cost of one KFAC factor build (ResNet-18, 512 rows): cProfile of the build after warm-up."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
kw = dict(fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=512)
def build():
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
    torch.cuda.synchronize()
for _ in range(3): build()
ts = []
for _ in range(5):
    t = time.perf_counter(); build(); ts.append(time.perf_counter() - t)
print(f"build: min {min(ts)*1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable(); build(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats("computers.py|_hip.py|streams.py|kfac_utils|kfac_math", 30)
