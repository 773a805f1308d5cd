"""float32 backward-data error of every ResNet-18 (CIFAR) convolution shape at batch 256 / 512 against float64 CPU."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(3, 64, 3, 1, 1, 32), (64, 64, 3, 1, 1, 32), (64, 128, 3, 2, 1, 32), (128, 128, 3, 1, 1, 16), (64, 128, 1, 2, 0, 32),
          (128, 256, 3, 2, 1, 16), (256, 256, 3, 1, 1, 8), (128, 256, 1, 2, 0, 16), (256, 512, 3, 2, 1, 8), (512, 512, 3, 1, 1, 4),
          (256, 512, 1, 2, 0, 8)]
for B in (256, 512):
    for (ci, co, k, s, p, hw) in shapes:
        conv64 = nn.Conv2d(ci, co, k, stride=s, padding=p, bias=False).double()
        conv32 = copy.deepcopy(conv64).float().to(dev)
        x64 = torch.randn(B, ci, hw, hw, dtype=torch.float64, requires_grad=True)
        x32 = x64.detach().float().to(dev).requires_grad_(True)
        y64 = conv64(x64); y32 = conv32(x32)
        g64 = torch.randn_like(y64); g32 = g64.float().to(dev)
        (dx64,) = torch.autograd.grad(y64, x64, g64); (dx32,) = torch.autograd.grad(y32, x32, g32)
        ef = float((y32.double().cpu() - y64).abs().max() / y64.abs().max())
        eb = float((dx32.double().cpu() - dx64).abs().max() / dx64.abs().max())
        print(f"B={B} conv {ci}->{co} k{k} s{s} {hw}x{hw}: forward {ef:.1e}  backward-data {eb:.1e}{'   <--' if max(ef, eb) > 1e-5 else ''}", flush=True)
