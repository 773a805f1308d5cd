import sys, time, torch
sys.path.insert(0, "/root/repo")
from benchmarks.models import ResNet18
import torch.nn as nn
dev = torch.device("cuda:0")
def run(flag, cl):
    torch.backends.cudnn.benchmark = flag
    torch.manual_seed(0)
    m = ResNet18(num_classes=10).to(dev).eval()
    if cl: m = m.to(memory_format=torch.channels_last)
    X = torch.rand(512, 3, 32, 32, device=dev); y = torch.randint(0, 10, (512,), device=dev)
    if cl: X = X.contiguous(memory_format=torch.channels_last)
    X.requires_grad_(True)
    for p in m.parameters(): p.requires_grad_(False)
    lf = nn.CrossEntropyLoss()
    def step():
        loss = lf(m(X), y); loss.backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); print(f"benchmark={flag} channels_last={cl}: {(time.perf_counter()-t)/20*1e3:.2f} ms", flush=True)
for flag in (False, True):
    for cl in (False, True):
        run(flag, cl)
