"""Graph-captured KFAC factor build vs the eager build: factors (same MC draws) and wall time.  ResNet-18 (C4), 512 rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import computers
from benchmarks.models import ResNet18, kfac_params

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
g = torch.Generator().manual_seed(4321)
X = torch.rand(rows, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (rows,), generator=g).to(dev)
X2 = torch.rand(rows, 3, 32, 32, generator=g).to(dev)
kw = dict(fisher_type="mc", mc_samples=1, separate_weight_and_bias=False, check_deterministic=False, num_data=rows)


def factors(K):
    _, B, _ = K
    out = []
    for blk in B:
        out.extend(list(blk))
    return out


def build(Xb, capture):
    computers._CAPTURE = capture
    return C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(Xb, y)], **kw)


def timed(fn, n=5):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best, r


build(X, False)
t_e, Ke = timed(lambda: build(X, False))
build(X, True)           # eager warm-up of the captured route
t0 = time.perf_counter(); build(X, True); torch.cuda.synchronize(); t_cap = 1e3 * (time.perf_counter() - t0)
t_c, Kc = timed(lambda: build(X, True))
fe, fc = factors(Ke), factors(Kc)
worst = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(fe, fc))
print(f"rows={rows}: eager {t_e:.2f} ms, capture call {t_cap:.1f} ms, replay {t_c:.2f} ms; max rel factor diff (same seed) {worst:.2e}", flush=True)
# live parameters / new data: replay on other data and after a .data update equals eager
Ke2, Kc2 = build(X2, False), build(X2, True)
w2 = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(factors(Ke2), factors(Kc2)))
for p in params.values():
    p.data.mul_(1.1)
Ke3, Kc3 = build(X2, False), build(X2, True)
w3 = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(factors(Ke3), factors(Kc3)))
d23 = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(factors(Ke2), factors(Ke3)))
print(f"other batch: {w2:.2e}; after p.data.mul_: {w3:.2e} (the update itself moved the factors by {d23:.2e}); "
      f"captured entries: {sum(isinstance(v, computers._CapturedBatch) for v in computers._CAPTURED.values())}", flush=True)
