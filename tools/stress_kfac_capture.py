"""Replays of the captured ResNet-18 factor build against the eager build, factor by factor (same MC draws): any factor that
deviates by more than 1e-4 is printed with its layer -- a race between the graph's branches would show up here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
import torch
from torch import nn
import curvlinops_amd as C
from curvlinops_amd import computers
from benchmarks.models import ResNet18, kfac_params

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
params = kfac_params(model)
g = torch.Generator().manual_seed(4321)
kw = dict(fisher_type="mc", mc_samples=1, separate_weight_and_bias=False, check_deterministic=False, num_data=512)
batches = [(torch.rand(512, 3, 32, 32, generator=g).to(dev), torch.randint(0, 10, (512,), generator=g).to(dev)) for _ in range(3)]


def build(b, capture):
    computers._CAPTURE = capture
    K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [batches[b]], **kw)
    names = [tuple(m.values())[0] for m in K._mapping] if hasattr(K, "_mapping") else None
    return [[f.clone() for f in blk] for blk in K[1]], names


for pix in (False, True):
    computers._PIXEL_GRAM = pix
    computers.reset_captured_builds()
    refs = [build(b, False)[0] for b in range(3)]
    again = [build(b, False)[0] for b in range(3)]
    e2e = max(float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)) for r, a in zip(refs, again) for br, ba in zip(r, a) for x, y in zip(br, ba))
    bad = 0
    worst = 0.0
    for it in range(30):
        b = it % 3
        fac, names = build(b, True)
        for li, (br, bc) in enumerate(zip(refs[b], fac)):
            for fi, (x, y) in enumerate(zip(br, bc)):
                d = float((x - y).abs().max() / x.abs().max().clamp_min(1e-30))
                worst = max(worst, d)
                if d > 1e-4:
                    bad += 1
                    print(f"  pixel={pix} iter {it} batch {b} block {li} factor {fi} order {x.shape[0]}: rel diff {d:.2e}", flush=True)
    print(f"pixel_gram={pix} queues={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: eager vs eager {e2e:.1e}; 30 captured builds: worst {worst:.1e}, {bad} factors beyond 1e-4", flush=True)
