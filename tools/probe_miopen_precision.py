"""Accuracy of float32 convolution forward / backward on this platform (MIOpen) against float64 on the CPU, with
torch.backends.cudnn.allow_tf32 on (PyTorch's default) and off, and the time of a ResNet-18 forward + backward."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from benchmarks.models import ResNet18
dev = torch.device("cuda:0")
print("defaults: cudnn.allow_tf32 =", torch.backends.cudnn.allow_tf32, " cuda.matmul.allow_tf32 =", torch.backends.cuda.matmul.allow_tf32)
torch.manual_seed(0)
cases = [(64, 64, 3, 1, 1, 32), (128, 256, 3, 2, 1, 16), (512, 512, 3, 1, 1, 4), (3, 64, 3, 1, 1, 32), (64, 128, 1, 2, 0, 32)]
for flag in (True, False):
    torch.backends.cudnn.allow_tf32 = flag
    for (ci, co, k, s, p, hw) in cases:
        conv64 = nn.Conv2d(ci, co, k, stride=s, padding=p, bias=False).double()
        conv32 = copy.deepcopy(conv64).float().to(dev)
        x64 = torch.randn(64, ci, hw, hw, dtype=torch.float64, requires_grad=True)
        x32 = x64.detach().float().to(dev).requires_grad_(True)
        y64 = conv64(x64); y32 = conv32(x32)
        g64 = torch.randn_like(y64); g32 = g64.float().to(dev)
        (dx64,) = torch.autograd.grad(y64, x64, g64); (dx32,) = torch.autograd.grad(y32, x32, g32)
        ef = float((y32.double().cpu() - y64).abs().max() / y64.abs().max())
        eb = float((dx32.double().cpu() - dx64).abs().max() / dx64.abs().max())
        print(f"allow_tf32={flag}: conv {ci}->{co} k{k} s{s} {hw}x{hw}: forward err {ef:.1e}  backward-data err {eb:.1e}", flush=True)
    model = ResNet18().to(dev).eval()
    for prm in model.parameters(): prm.requires_grad_(False)
    X = torch.rand(512, 3, 32, 32, device=dev, requires_grad=True); y = torch.randint(0, 10, (512,), device=dev)
    def step():
        nn.functional.cross_entropy(model(X), y).backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); print(f"allow_tf32={flag}: ResNet-18 forward + backward (512 rows): {(time.perf_counter()-t)/20*1e3:.2f} ms", flush=True)
