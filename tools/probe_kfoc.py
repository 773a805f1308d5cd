"""Scratch: KFOC build time (factored rearranged products on the GEMM engine + device Lanczos)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
import torch
from torch import nn
import curvlinops_amd as C
import models
dev = torch.device("cuda:0"); torch.manual_seed(0)
cases = {
    "mlp 784-512-256-10 B=256 type-2": (nn.Sequential(nn.Linear(784, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(), nn.Linear(256, 10)), (256, 784), "type-2", 1),
    "lenet B=256 type-2": (models.lenet5() if hasattr(models, "lenet5") else None, (256, 1, 32, 32), "type-2", 1),
    "lenet B=1024 mc": (models.lenet5() if hasattr(models, "lenet5") else None, (1024, 1, 32, 32), "mc", 1),
}
for name, (model, xs, fisher, mcs) in cases.items():
    if model is None: continue
    model = model.to(dev)
    X, y = torch.rand(*xs, device=dev), torch.randint(0, 10, (xs[0],), device=dev)
    params = dict(model.named_parameters())
    for sep in (False,):
        def build():
            return C.KFOCLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type=fisher, mc_samples=mcs,
                                        separate_weight_and_bias=sep, check_deterministic=False)
        K = build(); torch.cuda.synchronize()
        t0 = time.perf_counter(); K = build(); torch.cuda.synchronize(); t = time.perf_counter() - t0
        Kf = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type=fisher, mc_samples=mcs,
                                  separate_weight_and_bias=sep, check_deterministic=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Kf = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type=fisher, mc_samples=mcs,
                                  separate_weight_and_bias=sep, check_deterministic=False)
        torch.cuda.synchronize(); tk = time.perf_counter() - t0
        print(f"{name}: KFOC build {t*1e3:.1f} ms (KFAC build {tk*1e3:.1f} ms), D={K.shape[0]}")
