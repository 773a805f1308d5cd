"""ResNet-18 KFAC: all 42 damped Cholesky inverses for different numbers of inverse workers (linalg_native.concurrent_inverses),
optionally after an EKFAC basis build in the same process (the state the bench process is in)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import linalg_native as L
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18().to(dev).eval(); params = kfac_params(model)
X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False,
                         check_deterministic=False)
if os.environ.get("AFTER_EIGH"):
    C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False)
    torch.cuda.synchronize()
orig = L.concurrent_inverses
for w in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]:
    L.concurrent_inverses = lambda num_streams=2, distributed=False, w=w: orig(w, distributed)
    import curvlinops_amd.kfac as KF
    KF.linalg_native.concurrent_inverses = L.concurrent_inverses
    ts = []
    for i in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Kinv = K.inverse(damping=1e-3)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"queues {os.environ.get('GPU_MAX_HW_QUEUES', '4')} after_eigh {bool(os.environ.get('AFTER_EIGH'))} inverse workers {w}: " + " ".join(f"{t:.1f}" for t in ts) + " ms", flush=True)
