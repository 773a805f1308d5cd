"""Host-side profile (cProfile) of warm KFAC factor builds on ResNet-18 (C4): where do the ~2 ms above the plain
gradient + loss pass go?"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18(num_classes=10).to(dev).eval(); params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
def build():
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", mc_samples=1,
                             separate_weight_and_bias=False, check_deterministic=False)
    torch.cuda.synchronize(); return K
def plain():
    loss = nn.functional.cross_entropy(model(X), y); g = torch.autograd.grad(loss, list(params.values())); torch.cuda.synchronize()
for _ in range(3): build(); plain()
for name, fn in (("build", build), ("gradient + loss", plain)):
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name}: min {min(ts):.2f} median {sorted(ts)[3]:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): build()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
