import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvlinops_amd import _hip
_hip.load()
dev = torch.device("cuda:0")
n, b = 3072, 24
X = torch.randn(n + 8, n, device=dev); A = X.T @ X / n
mats = [A] * b
outs = [torch.empty(n, n, device=dev) for _ in range(b)]
status = torch.zeros(b, device=dev, dtype=torch.int32)
for _ in range(3): _hip.cholesky_inverse_batched_into(mats, [1e-3] * b, outs, status)
torch.cuda.synchronize()
