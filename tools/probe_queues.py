"""bench.py's ResNet-18 legs under a given GPU_MAX_HW_QUEUES: captured factor build, matvec, damped inverses, EKFAC bases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import linalg_native
from benchmarks.models import ResNet18, kfac_params

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
g = torch.Generator().manual_seed(4321)
X = torch.rand(512, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (512,), generator=g).to(dev)
kw = dict(fisher_type="mc", mc_samples=1, separate_weight_and_bias=False, check_deterministic=False, num_data=512)
def med(fn, n=5, warm=2):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2], min(ts), r
b_med, b_min, K = med(lambda: C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw), n=7, warm=3)
v = torch.rand(K.shape[1], device=dev)
m_med, m_min, _ = med(lambda: K @ v, n=10)
i_med, i_min, _ = med(lambda: K.inverse(damping=1e-3), n=5)
facs = [f for blk in K[1] for f in blk]
e_med, e_min, _ = med(lambda: linalg_native.eigh_many(facs), n=3, warm=1)
print(f"queues={sys.argv[1]}: build {b_med:.2f} (min {b_min:.2f}) ms | matvec {m_med:.2f} | inverses {i_med:.2f} (min {i_min:.2f}) | eigh_many {e_med:.1f} (min {e_min:.1f}) | f64 retries {linalg_native.FLOAT64_RETRIES}", flush=True)
