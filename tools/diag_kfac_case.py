"""Re-create one case of tools/fuzz_kfac.py: python tools/diag_kfac_case.py seed case"""
import os, sys, copy, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch import nn
import curvlinops_amd as C
import fuzz_kfac as F
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
warnings.simplefilter("ignore")
for case in range(target + 1):
    torch.manual_seed(1000 * seed + case)
    model64, shape, out = F.make_model(rng)
    model64 = model64.double()
    lossname = str(rng.choice(["mse", "ce"])); red = str(rng.choice(["mean", "sum"]))
    scale = 10.0 ** rng.uniform(-2, 2)
    data64 = []
    for _ in range(int(rng.integers(1, 3))):
        n = int(rng.integers(2, 9))
        X = torch.rand(n, *shape, dtype=torch.float64) * scale
        y = torch.randint(0, out, (n,)) if lossname == "ce" else torch.rand(n, out, dtype=torch.float64)
        data64.append((X, y))
    kw = dict(fisher_type=str(rng.choice(["empirical", "type-2"])), kfac_approx=str(rng.choice(["expand", "reduce"])),
              separate_weight_and_bias=bool(rng.random() < 0.5), check_deterministic=False)
    if case < target:   # consume the rng exactly as run() does
        D = sum(p.numel() for p in model64.parameters())
        has_conv = any(isinstance(m, nn.Conv2d) for m in model64)
        for cls in (0, 1):
            if cls == 1 and has_conv and kw["kfac_approx"] == "reduce":
                continue
            torch.rand(D, 3, dtype=torch.float64)
print([type(m).__name__ for m in model64], shape, lossname, red, scale, kw)
loss = (nn.MSELoss if lossname == "mse" else nn.CrossEntropyLoss)(reduction=red)
model32 = copy.deepcopy(model64).float().to(dev)
data32 = [(X.float().to(dev), y.to(dev) if y.dtype == torch.int64 else y.float().to(dev)) for X, y in data64]
K64 = C.KFACLinearOperator(model64, loss, dict(model64.named_parameters()), data64, **kw)
K32 = C.KFACLinearOperator(model32, loss, dict(model32.named_parameters()), data32, **kw)
D = K64.shape[1]
V = torch.rand(D, 3, dtype=torch.float64) - 0.5
ref = K64 @ V
damp = float(os.environ.get("DAMP", 1e-2 * float(ref.abs().max()) / float(V.abs().max())))
print("damping", damp, "|K V|", float(ref.abs().max()))
P, Kb, PT = K64
P32, Kb32, _ = K32
for b64, b32 in zip(Kb, Kb32):
    fs64 = getattr(b64, "_factors", None) or getattr(b64, "factors", None)
    print(type(b64).__name__, [tuple(f.shape) for f in (fs64 or [])], [float(f.abs().max()) for f in (fs64 or [])])
for mode in ({"use_exact_damping": True}, {"use_heuristic_damping": True}, {}):
    try:
        i64 = K64.inverse(damping=damp, **mode) @ V
        i32 = K32.inverse(damping=damp, **mode) @ V.float().to(dev)
        print(mode, "rel err", F.rel(i32, i64), "nan32", bool(torch.isnan(i32).any()), "nan64", bool(torch.isnan(i64).any()))
        # per parameter block error
        off = 0
        for name, prm in model64.named_parameters():
            sl = slice(off, off + prm.numel()); off += prm.numel()
            print("   ", name, tuple(prm.shape), "err", float((i32[sl].double().cpu() - i64[sl]).abs().max() / i64[sl].abs().max().clamp_min(1e-300)), "|i64|", float(i64[sl].abs().max()))
    except Exception as e:
        print(mode, "exception", type(e).__name__, str(e)[:120])

print("---- block-level check of the exact-damping inverse")
for bi, (b64, b32) in enumerate(zip(Kb, Kb32)):
    f64 = list(b64); f32 = list(b32)
    n = 1
    for f in f64: n *= f.shape[0]
    if n > 5000: continue
    x = torch.rand(n, 3, dtype=torch.float64) - 0.5
    def dense(fs):
        M = fs[0].double().cpu().contiguous()
        for f in fs[1:]: M = torch.kron(M, f.double().cpu().contiguous())
        return M
    M64, M32 = dense(f64), dense(f32)
    I = torch.eye(n, dtype=torch.float64)
    r64 = torch.linalg.solve(M64 + damp * I, x)
    r32 = torch.linalg.solve(M32 + damp * I, x)
    g32 = (b32.inverse(damping=damp, use_exact_damping=True) @ x.float().to(dev)).double().cpu()
    g64 = b64.inverse(damping=damp, use_exact_damping=True) @ x
    sc = float(r64.abs().max())
    print(f"block {bi} {[tuple(f.shape) for f in f64]}: fp32 kernels vs dense(fp32 factors) {float((g32-r32).abs().max())/sc:.1e} | "
          f"dense(fp32 factors) vs dense(fp64 factors) {float((r32-r64).abs().max())/sc:.1e} | fp64 path vs dense {float((g64-r64).abs().max())/sc:.1e} | "
          f"factor diff {[float((a.double().cpu()-b).abs().max()/b.abs().max()) for a,b in zip(f32,f64)]}")

from curvlinops_amd import linalg_native as L
for bi, (b64, b32) in enumerate(zip(Kb, Kb32)):
    for f64, f32 in zip(list(b64), list(b32)):
        if f64.shape[0] < 100: continue
        l64 = torch.linalg.eigvalsh(f64)
        l32, Q32 = L.eigh(f32)
        print(f"block {bi} factor {tuple(f64.shape)}: symmetric? {float((f32 - f32.T).abs().max()):.1e}  fp64 eig max {float(l64.max()):.3e} #>1e-10: {int((l64 > 1e-10 * l64.max()).sum())}"
              f" | fp32 eig max {float(l32.max()):.3e} min {float(l32.min()):.3e} | max |l32 - l64| {float((l32.double().cpu() - l64).abs().max()):.2e}")
        Q = Q32.double().cpu(); A32 = f32.double().cpu()
        print("   residual", float((A32 @ Q - Q * l32.double().cpu()).abs().max() / A32.abs().max()), "orth", float((Q.T @ Q - torch.eye(Q.shape[0], dtype=torch.float64)).abs().max()),
              "strides", Q32.stride(), "contig", Q32.is_contiguous())

print("---- solvers on the large factors of this case")
os.makedirs(os.path.join(os.environ.get("GRAFT_REPO_ROOT", ".."), "gpurun_out"), exist_ok=True)
for bi, b32 in enumerate(Kb32):
    for f32 in list(b32):
        n = f32.shape[0]
        if n < 100: continue
        torch.save(f32.cpu(), os.path.join(os.environ.get("GRAFT_REPO_ROOT", ".."), "gpurun_out", f"factor_case_{seed}_{target}_{bi}.pt"))
        A32 = f32.double().cpu(); I = torch.eye(n, dtype=torch.float64)
        print("  unique rows:", len(torch.unique(f32.cpu(), dim=0)), "zero rows:", int((f32.abs().sum(1) == 0).sum()), "max", float(f32.abs().max()), "diag min", float(f32.diag().min()))
        for name, fn in (("torch raw", lambda M: tuple(torch.linalg.eigh(M))), ("torch normalised", L._torch_eigh_scaled), ("sytrd", L.eigh_sytrd),
                         ("torch on CPU fp32", lambda M: tuple(torch.linalg.eigh(M.cpu())))):
            lam, Q = fn(f32)
            Q64, l64 = Q.double().cpu(), lam.double().cpu()
            print(f"   {name}: orth {float((Q64.T @ Q64 - I).abs().max()):.1e} res {float((A32 @ Q64 - Q64 * l64).abs().max() / A32.abs().max()):.1e}")
