"""Where does the float32-vs-float64 difference of ResNet-18's gradient covariances come from?  One float32
forward / backward with tensor hooks on every conv / linear output; G_l recomputed in float64 from the SAME float32
output-gradients vs the product's factors (isolates our SYRK kernels from upstream differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import curvlinops_amd as C
from benchmarks.models import ResNet18, kfac_params
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
B = 512
X, y = torch.rand(B, 3, 32, 32, device=dev), torch.randint(0, 10, (B,), device=dev)
lf = nn.CrossEntropyLoss()
K = C.KFACLinearOperator(model, lf, kfac_params(model), [(X, y)], fisher_type="empirical", separate_weight_and_bias=False,
                         check_deterministic=False)
facs = [S for blk in K[1] for S in blk]
mods = [m for m in model.modules() if isinstance(m, (nn.Conv2d, nn.Linear))]
grads = {}
def make_hook(i):
    def fwd(mod, inp, out):
        out.register_hook(lambda g: grads.__setitem__(i, g.detach()))
    return fwd
hs = [m.register_forward_hook(make_hook(i)) for i, m in enumerate(mods)]
Xr = X.clone().requires_grad_(True)
loss = lf(model(Xr), y); loss.backward()
for h in hs: h.remove()
print("layer  d_out  rows | product G vs float64 SYRK of the same float32 gradients")
for i, m in enumerate(mods):
    g = grads[i]
    g2 = g.movedim(1, -1).reshape(-1, g.shape[1]).double() if g.dim() == 4 else g.double()
    ref = g2.T @ g2 * (B * B / B)      # mean reduction: (B T)^2 / (T N) with T = 1, N = B
    G = facs[2 * i]
    print(f"{i:3d} {g2.shape[1]:5d} {g2.shape[0]:7d} | {float((G.double() - ref).abs().max() / ref.abs().max()):.1e}")

print("---- run-to-run: two more float32 passes of the same model / data, output-gradients compared per layer")
def one_pass():
    got = {}
    def mk(i):
        def fwd(mod, inp, out):
            out.register_hook(lambda g: got.__setitem__(i, g.detach().clone()))
        return fwd
    hs = [m.register_forward_hook(mk(i)) for i, m in enumerate(mods)]
    Xr = X.clone().requires_grad_(True)
    out = model(Xr)
    lf(out, y).backward()
    for h in hs: h.remove()
    return got, out.detach().clone()
(g1, o1), (g2, o2) = one_pass(), one_pass()
print("forward outputs identical:", bool(torch.equal(o1, o2)))
for i in range(len(mods)):
    d = float((g1[i] - g2[i]).abs().max() / g1[i].abs().max())
    d0 = float((g1[i] - grads[i]).abs().max() / g1[i].abs().max())
    print(f"{i:3d}: pass A vs pass B {d:.1e} | pass A vs first pass {d0:.1e}")

print("---- what the product's gradient hook sees vs the manual pass")
from curvlinops_amd import computers as CM
seen = []
orig = CM.HipKFACComputer._grad_hook
def spy(self, grad_output, group, hyper, store, stacked=False):
    seen.append((tuple(group.values()), grad_output.detach().clone()))
    return orig(self, grad_output, group, hyper, store, stacked)
CM.HipKFACComputer._grad_hook = spy
K2 = C.KFACLinearOperator(model, lf, kfac_params(model), [(X, y)], fisher_type="empirical", separate_weight_and_bias=False,
                          check_deterministic=False)
CM.HipKFACComputer._grad_hook = orig
names = [n for n, m in model.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))]
byname = {k[0].rsplit(".", 1)[0]: g for k, g in seen}
facs2 = [S for blk in K2[1] for S in blk]
for i, nme in enumerate(names):
    g = byname[nme]
    d = float((g - g1[i]).abs().max() / g1[i].abs().max())
    g2 = g.movedim(1, -1).reshape(-1, g.shape[1]).double() if g.dim() == 4 else g.double()
    ref = g2.T @ g2 * B
    e = float((facs2[2 * i].double() - ref).abs().max() / ref.abs().max())
    print(f"{i:3d} {nme:28s}: hook gradient vs manual {d:.1e} | product G vs float64 SYRK of ITS OWN gradient {e:.1e}")
