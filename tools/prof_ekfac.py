import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd.computers import HipEKFACComputer, _use_params
from curvlinops_amd import linalg_native
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = ResNet18().to(dev).eval(); params = kfac_params(model); B = 512
X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
comp = HipEKFACComputer(model, nn.CrossEntropyLoss(), params, [(X, y)], fisher_type="mc", separate_weight_and_bias=False,
                        check_deterministic=False, num_data=B)
with _use_params(comp._model_module, comp._params):
    A, G, mapping = comp._compute_kronecker_factors()
    Qa = {k: linalg_native.eigh(v)[1] for k, v in A.items()}
    Qg = {k: linalg_native.eigh(v)[1] for k, v in G.items()}
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lam = comp._eigenvalue_correction(Qa, Qg, mapping)
        torch.cuda.synchronize(); print(f"correction pass {i}: {(time.perf_counter()-t0)*1e3:.1f} ms")
