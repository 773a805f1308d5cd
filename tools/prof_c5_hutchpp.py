"""One warm hutchpp_trace(EF, 96) on C5 (12-layer d = 768 encoder, D = 85 M, 8 x 128 tokens) between two marker launches,
for rocprofv3 --kernel-trace: who owns the 0.83 s -- clo:: kernels or the framework's (torch.func products)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import _hip
from benchmarks.models import Encoder

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = Encoder().to(dev).eval()
p5 = dict(enc.named_parameters())
X5, y5 = torch.rand(8, 128, 768, device=dev), torch.randint(0, 10, (8,), device=dev)
EF = C.EFLinearOperator(enc, nn.CrossEntropyLoss(), p5, [(X5, y5)], check_deterministic=False, num_data=8)
C.hutchpp_trace(EF, num_matvecs=96)
torch.cuda.synchronize()
mark = torch.zeros(4099, device=dev)
_hip.axpby(mark, mark, 1.0, 0.0)
torch.cuda.synchronize()
C.hutchpp_trace(EF, num_matvecs=96)
torch.cuda.synchronize()
_hip.axpby(mark, mark, 1.0, 0.0)
torch.cuda.synchronize()
