"""bench.py's kfac leg, build by build (ResNet-18 C4): eager vs captured, with the pixel-Gram route on / off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "16")
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import computers
from benchmarks.models import ResNet18, kfac_params

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
g = torch.Generator().manual_seed(4321)
X = torch.rand(512, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (512,), generator=g).to(dev)
kw = dict(fisher_type="mc", mc_samples=1, separate_weight_and_bias=False, check_deterministic=False, num_data=512)
ref = None
for pix in (False, True):
    for cap in (False, True):
        computers._PIXEL_GRAM, computers._CAPTURE = pix, cap
        computers.reset_captured_builds()
        times = []
        for i in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
            torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
        fac = [f for blk in K[1] for f in blk]
        if ref is None:
            ref = fac
        worst = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(ref, fac))
        print(f"pixel_gram={pix} capture={cap}: " + " ".join(f"{t:.2f}" for t in times) + f" ms | max rel diff vs first config {worst:.1e}", flush=True)
v = torch.rand(K.shape[1], device=dev)
for grp in (False, True):
    K[1].assume_frozen = grp
    K @ v; torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        K @ v
    torch.cuda.synchronize(); print(f"kfac matvec (assume_frozen={grp}): {1e2*(time.perf_counter()-t0):.3f} ms", flush=True)
