"""Replays cases of tools/fuzz_native.py (same seed stream) and compares the native fp32 product and the torch fp32 path
against the torch path in float64: python tools/diag_fuzz_case.py seed case [case ...]"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch import nn
import curvlinops_amd as C
import fuzz_native as F
F.WIDE = float(os.environ.get("CLO_FUZZ_WIDE", "0.3"))
seed, targets = int(sys.argv[1]), {int(a) for a in sys.argv[2:]}
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
for case in range(max(targets) + 1):
    state = rng.bit_generator.state
    if case not in targets:
        # consume the stream exactly as the fuzzer does, on tiny stand-ins: re-run the generator logic without the products
        orig = [C.GGNLinearOperator, C.EFLinearOperator, C.HessianLinearOperator]
        class Skip:
            def __init__(self, *a, **k): self.uses_native_kernels = True; self.shape = (1, 1); self._native = None
            def __matmul__(self, v): return torch.zeros(1, v.shape[1] if v.dim() > 1 else 1, device=dev).squeeze(-1) if v.dim() == 1 else torch.zeros(1, v.shape[1], device=dev)
        C.GGNLinearOperator = C.EFLinearOperator = C.HessianLinearOperator = Skip
        try:
            F._one_case(case, rng, dev, [])
        finally:
            C.GGNLinearOperator, C.EFLinearOperator, C.HessianLinearOperator = orig
        continue
    # the target case: rebuild it by hand from the same draws
    L = int(rng.integers(1, 5)); align = rng.random() < 0.5; wide = rng.random() < F.WIDE
    dims = [int(rng.integers(4, 400)) * 4 for _ in range(L + 1)] if wide else [int(rng.integers(1, 40)) * 4 if align else int(rng.integers(2, 150)) for _ in range(L + 1)]
    if rng.random() < 0.5: dims[-1] = int(rng.integers(1, 17))
    bias = bool(rng.random() < 0.8)
    layers, acts = [], []
    for l in range(L):
        layers.append(nn.Linear(dims[l], dims[l + 1], bias=bias))
        act = F.ACTS[int(rng.integers(0, 4))] if l < L - 1 else None
        acts.append(None if act is None else act.__name__)
        if act is not None: layers.append(act())
    torch.manual_seed(case)
    model = nn.Sequential(*layers).to(dev); params = dict(model.named_parameters())
    lossname = ["mse", "ce", "bce"][int(rng.integers(0, 3))]; red = ["mean", "sum"][int(rng.integers(0, 2))]
    loss = F.LOSSES[lossname](reduction=red)
    data = []
    for _ in range(int(rng.integers(1, 4))):
        N = int(rng.choice([1, 3, 8, 9, 13, 16, 17, 24, 31, 32, 33, 64, 70]))
        X = torch.rand(N, dims[0], device=dev) - 0.5
        y = torch.randint(0, dims[-1], (N,), device=dev) if lossname == "ce" else torch.rand(N, dims[-1], device=dev)
        data.append((X, y))
    m64 = copy.deepcopy(model).double(); p64 = dict(m64.named_parameters())
    d64 = [(X.double(), y if lossname == "ce" else y.double()) for X, y in data]
    print(f"case {case}: dims={dims} acts={acts} bias={bias} loss={lossname}/{red} Ns={[x.shape[0] for x, _ in data]}")
    with torch.no_grad():
        out = model(data[0][0]); print("  |logits| max", float(out.abs().max()))
    for cls in (C.GGNLinearOperator, C.EFLinearOperator, C.HessianLinearOperator):
        nat = cls(model, loss, params, data, check_deterministic=False)
        ref = cls(model, loss, params, data, check_deterministic=False); ref._native = None
        r64 = cls(m64, loss, p64, d64, check_deterministic=False)
        D = nat.shape[1]
        for K in (1, int(rng.choice([3, 8, 12]))):
            V = torch.rand(D, K, device=dev) - 0.5
            a = (nat @ V[:, 0].contiguous()).unsqueeze(1) if K == 1 else nat @ V
            b = ref @ V
            c = r64 @ V.double()
            sc = c.abs().max()
            print(f"  {cls.__name__} K={K}: native vs f64 {float((a.double() - c).abs().max() / sc):.2e}, torch f32 vs f64 {float((b.double() - c).abs().max() / sc):.2e}, native vs torch f32 {float((a - b).abs().max() / b.abs().max()):.2e}")

    if os.environ.get("PER_BATCH"):
        for bi, (dd, dd64) in enumerate(zip(data, d64)):
            for cls in (C.GGNLinearOperator, C.HessianLinearOperator):
                nat = cls(model, loss, params, [dd], check_deterministic=False)
                r64 = cls(m64, loss, p64, [dd64], check_deterministic=False)
                v = torch.rand(nat.shape[1], device=dev) - 0.5
                a, c = nat @ v, r64 @ v.double()
                # per-parameter-block errors
                errs, off = [], 0
                for n_, p_ in params.items():
                    k = p_.numel(); errs.append(f"{n_}:{float((a[off:off+k].double() - c[off:off+k]).abs().max() / c.abs().max()):.1e}"); off += k
                print(f"  batch {bi} (N={dd[0].shape[0]}) {cls.__name__}: {float((a.double() - c).abs().max() / c.abs().max()):.2e}  " + " ".join(errs))

    if os.environ.get("PER_BATCH"):
        for rows in (33, 64, 65, 72, 128):
            Xc = torch.rand(rows, dims[0], device=dev) - 0.5
            yc = torch.randint(0, dims[-1], (rows,), device=dev) if lossname == "ce" else torch.rand(rows, dims[-1], device=dev)
            for cls in (C.GGNLinearOperator, C.HessianLinearOperator):
                nat = cls(model, loss, params, [(Xc, yc)], check_deterministic=False)
                r64 = cls(m64, loss, p64, [(Xc.double(), yc if lossname == "ce" else yc.double())], check_deterministic=False)
                v = torch.rand(nat.shape[1], device=dev) - 0.5
                a, c = nat @ v, r64 @ v.double()
                errs, off = [], 0
                for n_, p_ in params.items():
                    k = p_.numel(); errs.append(f"{n_}:{float((a[off:off+k].double() - c[off:off+k]).abs().max() / c.abs().max()):.1e}"); off += k
                print(f"  one batch of {rows} rows {cls.__name__}: {float((a.double() - c).abs().max() / c.abs().max()):.2e}  " + " ".join(errs))

    if os.environ.get("PER_BATCH"):
        for sub in ([0, 1], [0, 2], [2, 0], [0, 1, 2]):
            dsub, dsub64 = [data[i] for i in sub], [d64[i] for i in sub]
            for cls in (C.GGNLinearOperator, C.HessianLinearOperator):
                nat = cls(model, loss, params, dsub, check_deterministic=False)
                r64 = cls(m64, loss, p64, dsub64, check_deterministic=False)
                v = torch.rand(nat.shape[1], device=dev) - 0.5
                a, c = nat @ v, r64 @ v.double()
                errs, off = [], 0
                for n_, p_ in params.items():
                    k = p_.numel(); errs.append(f"{n_}:{float((a[off:off+k].double() - c[off:off+k]).abs().max() / c[off:off+k].abs().max()):.1e}"); off += k
                print(f"  batches {sub} {cls.__name__}: {float((a.double() - c).abs().max() / c.abs().max()):.2e}  (per block, relative to the block) " + " ".join(errs))

    if os.environ.get("PER_BATCH"):
        Xc = torch.cat([d[0] for d in data]); yc = torch.cat([d[1] for d in data])
        for tag, dd, dd64 in (("hand-merged single batch", [(Xc, yc)], [(Xc.double(), yc if lossname == "ce" else yc.double())]),
                              ("three batches, merging off", data, d64), ("three batches, merging on", data, d64)):
            for cls in (C.GGNLinearOperator, C.HessianLinearOperator):
                nat = cls(model, loss, params, dd, check_deterministic=False)
                if "off" in tag: nat._MERGE_MAX_ROWS = 0
                r64 = cls(m64, loss, p64, dd64, check_deterministic=False)
                for rep in range(3):
                    v = torch.rand(nat.shape[1], device=dev) - 0.5
                    a, c = nat @ v, r64 @ v.double()
                    print(f"  {tag} {cls.__name__} rep {rep}: {float((a.double() - c).abs().max() / c.abs().max()):.2e}")

    if os.environ.get("PER_BATCH"):
        print("  --- determinism / data dependence")
        Xc = torch.cat([d[0] for d in data]); yc = torch.cat([d[1] for d in data])
        Xr = torch.rand(72, dims[0], device=dev) - 0.5; yr = torch.randint(0, dims[-1], (72,), device=dev)
        for tag, XX, yy in (("cat data", Xc, yc), ("fresh random", Xr, yr), ("cat X, random y", Xc, yr), ("random X, cat y", Xr, yc), ("cat data, rows reversed", Xc.flip(0).contiguous(), yc.flip(0).contiguous())):
            nat = C.GGNLinearOperator(model, loss, params, [(XX, yy)], check_deterministic=False)
            r64 = C.GGNLinearOperator(m64, loss, p64, [(XX.double(), yy)], check_deterministic=False)
            v = torch.rand(nat.shape[1], device=dev) - 0.5
            a1 = nat @ v; a2 = nat @ v; c = r64 @ v.double()
            print(f"  {tag}: vs f64 {float((a1.double() - c).abs().max() / c.abs().max()):.2e}, second call equal to first: {bool(torch.equal(a1, a2))}, labels unique {int(yy.unique().numel())}")

    if os.environ.get("PER_BATCH"):
        print("  --- jacobians on the cat data")
        Xc = torch.cat([d[0] for d in data])
        for XX, tag in ((Xc, "cat X"), (torch.rand(72, dims[0], device=dev) - 0.5, "random X")):
            J = C.JacobianLinearOperator(model, params, [(XX, None)] if False else [(XX, torch.zeros(72, device=dev))], check_deterministic=False)
            J64 = C.JacobianLinearOperator(m64, p64, [(XX.double(), torch.zeros(72, device=dev))], check_deterministic=False)
            v = torch.rand(J.shape[1], device=dev) - 0.5
            a, c = J @ v, J64 @ v.double()
            a = a.reshape(72, -1); c = c.reshape(72, -1)
            rowerr = (a.double() - c).abs().amax(1) / c.abs().max()
            print(f"  JVP {tag}: {float(rowerr.max()):.2e}; worst rows {rowerr.topk(4).indices.tolist()} {[f'{x:.1e}' for x in rowerr.topk(4).values.tolist()]}  native={J.uses_native_kernels}")
            JT = J.adjoint(); JT64 = J64.adjoint()
            u = torch.rand(JT.shape[1], device=dev) - 0.5
            a, c = JT @ u, JT64 @ u.double()
            errs, off = [], 0
            for n_, p_ in params.items():
                k = p_.numel(); errs.append(f"{n_}:{float((a[off:off+k].double() - c[off:off+k]).abs().max() / c[off:off+k].abs().max()):.1e}"); off += k
            print(f"  VJP {tag}: {float((a.double() - c).abs().max() / c.abs().max()):.2e}  " + " ".join(errs))

    if os.environ.get("PER_BATCH"):
        with torch.no_grad():
            Xc = torch.cat([d[0] for d in data])
            z64 = m64[0](Xc.double()); z32 = model[0](Xc)
            k = z64.abs().argmin(); r, j = int(k // z64.shape[1]), int(k % z64.shape[1])
            print(f"  smallest |z_1| in float64: {float(z64.flatten()[k]):.3e} at row {r}, feature {j}; float32 torch value there {float(z32[r, j]):.3e}")
