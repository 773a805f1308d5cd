import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import Encoder
dev = torch.device("cuda:0"); torch.manual_seed(0)
step = sys.argv[1]
D = 85_000_000
if step == "qr":
    X = torch.randn(D, 32, device=dev)
    print("qr start", flush=True); Q, R = torch.linalg.qr(X); torch.cuda.synchronize(); print("qr ok", Q.shape, flush=True)
elif step == "mm":
    Q = torch.randn(D, 32, device=dev); G = torch.randn(D, 32, device=dev)
    print("QtG", flush=True); T = Q.T @ G; torch.cuda.synchronize(); print("ok", flush=True)
    print("Q T", flush=True); Y = Q @ T; torch.cuda.synchronize(); print("ok", flush=True)
    print(torch.einsum("ij,ij", G, Y).item())
elif step == "probes":
    from curvlinops_amd.trace import random_matrix
    S = random_matrix(D, 32, "rademacher", dev, torch.float32); torch.cuda.synchronize(); print("probes ok", S.abs().mean().item())
else:
    model = Encoder(layers=int(sys.argv[2]) if len(sys.argv) > 2 else 12).to(dev).eval()
    params = dict(model.named_parameters())
    B = 8
    X, y = torch.rand(B, 128, 768, device=dev), torch.randint(0, 10, (B,), device=dev)
    op = C.EFLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], check_deterministic=False, num_data=B)
    D = op.shape[1]; print("D", D, flush=True)
    v = torch.rand(D, device=dev)
    t0 = time.perf_counter(); r = op @ v; torch.cuda.synchronize(); print("matvec ok", time.perf_counter() - t0, flush=True)
    V = torch.rand(D, 32, device=dev)
    t0 = time.perf_counter(); r = op @ V; torch.cuda.synchronize(); print("matmat ok", time.perf_counter() - t0, flush=True)
