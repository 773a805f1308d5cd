"""Which of bench.py's secondary legs slows a KFAC build captured after it?  argv: any of c3 c4kfac c4eigh c4ekfac c5 empty"""
import os, sys, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import computers, linalg_native
from benchmarks.models import Encoder, ResNet18, kfac_params, lenet5
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
for what in sys.argv[1:]:
    if what == "c3":
        net3 = lenet5().to(dev); p3 = dict(net3.named_parameters())
        X3, y3 = torch.rand(1024, 1, 32, 32, device=dev), torch.randint(0, 10, (1024,), device=dev)
        for ft in ("mc", "type-2"):
            for _ in range(3): K3 = C.KFACLinearOperator(net3, nn.CrossEntropyLoss(), p3, [(X3, y3)], fisher_type=ft, separate_weight_and_bias=False, check_deterministic=False, num_data=1024)
        K3.inverse(damping=1e-3)
    if what.startswith("c4"):
        model = ResNet18().to(dev).eval(); params = kfac_params(model)
        X, y = torch.rand(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
        kw = dict(fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=512)
        if what == "c4kfac":
            for _ in range(3): K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
        if what == "c4eigh":
            K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
            facs = [S for blk in K[1] for S in blk]
            for _ in range(3): linalg_native.eigh_many(facs)
        if what == "c4ekfac":
            for _ in range(3): E = C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
    if what == "c5":
        enc = Encoder().to(dev).eval(); p5 = dict(enc.named_parameters())
        X5, y5 = torch.rand(8, 128, 768, device=dev), torch.randint(0, 10, (8,), device=dev)
        EF = C.EFLinearOperator(enc, nn.CrossEntropyLoss(), p5, [(X5, y5)], check_deterministic=False, num_data=8)
        C.hutchpp_trace(EF, num_matvecs=96)
    if what == "empty":
        gc.collect(); torch.cuda.empty_cache()
    torch.cuda.synchronize()
out = bench.kfac_leg(dev, 1, 0)
print(f"after {sys.argv[1:]}: build {out['ms_per_batch']:.2f} ms, captured builds {len(computers._CAPTURED)}", flush=True)
