"""Randomised GPU fp32 vs CPU float64 comparison of the KFAC / EKFAC operators of this package on random small
nets (Linear / Conv2d stacks, strides, paddings, with and without bias), losses, Fisher types (empirical, type-2),
expand / reduce, joint or separate weight + bias, input scales 1e-2 ... 1e2: K @ V, K^-1 @ V (damped) and E @ V.
The CPU float64 path is pinned to the reference by tests/golden.    python tools/fuzz_kfac.py [seed] [cases]"""
import os, sys, copy, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch import nn
import curvlinops_amd as C


class MeanPool(nn.Module):
    """[B, T, d] -> [B, d]: ends the weight-sharing part of a sequence model."""

    def forward(self, x):
        return x.mean(dim=1)


BIG = os.environ.get("CLO_FUZZ_BIG") is not None   # wider layers / more rows: more kernel paths (split-K, tall-skinny Gram)


def make_model(rng):
    kind = rng.random()
    conv = kind < 0.4
    layers = []
    if kind >= 0.8:   # sequence model: Linear layers applied to [B, T, d] (weight sharing over T), then pooled
        T, d = int(rng.integers(2, 6)), int(rng.integers(2, 8))
        shape = (T, d)
        for _ in range(int(rng.integers(1, 3))):
            do = int(rng.integers(2, 8))
            layers += [nn.Linear(d, do, bias=bool(rng.random() < 0.8)), nn.Tanh() if rng.random() < 0.5 else nn.ReLU()]
            d = do
        layers.append(MeanPool())
    elif conv:
        c, h = int(rng.integers(1, 12 if BIG else 4)), int(rng.integers(6, 20 if BIG else 12))
        shape = (c, h, h)
        for _ in range(int(rng.integers(1, 3))):
            co = int(rng.integers(2, 40 if BIG else 6))
            k = int(rng.integers(1, 4))
            s, p = int(rng.integers(1, 3)), int(rng.integers(0, 2))
            if (h + 2 * p - k) // s + 1 < 1:
                break
            layers += [nn.Conv2d(c, co, k, stride=s, padding=p, bias=bool(rng.random() < 0.8)), nn.ReLU() if rng.random() < 0.5 else nn.Tanh()]
            c, h = co, (h + 2 * p - k) // s + 1
        layers.append(nn.Flatten())
        d = c * h * h
    else:
        d = int(rng.integers(2, 300 if BIG else 12))
        shape = (d,)
    for _ in range(int(rng.integers(0, 3))):
        do = int(rng.integers(2, 200 if BIG else 10))
        layers += [nn.Linear(d, do, bias=bool(rng.random() < 0.8)), nn.Sigmoid() if rng.random() < 0.5 else nn.ReLU()]
        d = do
    out = int(rng.integers(2, 6))
    layers.append(nn.Linear(d, out, bias=bool(rng.random() < 0.8)))
    return nn.Sequential(*layers), shape, out


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def run(seed, ncase):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst, fails = 0.0, []
    for case in range(ncase):
        torch.manual_seed(1000 * seed + case)
        model64, shape, out = make_model(rng)
        model64 = model64.double()
        model32 = copy.deepcopy(model64).float().to(dev)
        lossname = str(rng.choice(["mse", "ce", "bce"]))
        red = str(rng.choice(["mean", "sum"]))
        loss = {"mse": nn.MSELoss, "ce": nn.CrossEntropyLoss, "bce": nn.BCEWithLogitsLoss}[lossname](reduction=red)
        scale = 10.0 ** rng.uniform(-2, 2)
        data64 = []
        for _ in range(int(rng.integers(1, 3))):
            n = int(rng.integers(2, 70 if BIG else 9))
            X = torch.rand(n, *shape, dtype=torch.float64) * scale
            y = (torch.randint(0, out, (n,)) if lossname == "ce" else
                 torch.randint(0, 2, (n, out)).double() if lossname == "bce" else torch.rand(n, out, dtype=torch.float64))
            data64.append((X, y))
        data32 = [(X.float().to(dev), y.to(dev) if y.dtype == torch.int64 else y.float().to(dev)) for X, y in data64]
        kw = dict(fisher_type=str(rng.choice(["empirical", "type-2"])), kfac_approx=str(rng.choice(["expand", "reduce"])),
                  separate_weight_and_bias=bool(rng.random() < 0.5), check_deterministic=False)
        what = f"case {case}: {[type(m).__name__ for m in model64]} shape {shape} loss {lossname}/{red} scale {scale:.1e} {kw}"
        try:
            has_conv = any(isinstance(m, (nn.Conv2d, MeanPool)) for m in model64)   # any weight sharing
            for cls in (C.KFACLinearOperator, C.EKFACLinearOperator):
                if (cls is C.EKFACLinearOperator and has_conv and kw["kfac_approx"] == "reduce"
                        and os.environ.get("CLO_FUZZ_NOSKIP") is None):
                    # the eigenvalues are re-fitted on EXPAND-format patches (ekfac_hooks.py:435-440) in the
                    # eigenbasis of the REDUCE-format covariance, whose null space (few rows here) has no
                    # distinguished basis: the result is not unique, fp32 and fp64 legitimately differ
                    continue
                p64 = dict(model64.named_parameters())
                p32 = dict(model32.named_parameters())
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    K64 = cls(model64, loss, p64, data64, **kw)
                    K32 = cls(model32, loss, p32, data32, **kw)
                D = K64.shape[1]
                V = torch.rand(D, 3, dtype=torch.float64) - 0.5
                ref = K64 @ V
                got = K32 @ V.float().to(dev)
                e = rel(got, ref)
                worst = max(worst, e)
                if not e < 2e-3:
                    fails.append(f"{what}: {cls.__name__} @ V err {e:.1e}")
                    if os.environ.get("CLO_FUZZ_DEBUG"):
                        Kd = C.KFACLinearOperator(model64, loss, p64, data64, **kw)
                        print(what, "rows per batch", [x.shape[0] for x, _ in data64])
                        for block in Kd[1]:
                            for f in block:
                                ev = torch.linalg.eigvalsh(f)
                                print("   factor", tuple(f.shape), "eigenvalues / max:", [f"{float(v):.2e}" for v in (ev / ev.abs().max().clamp_min(1e-300))])
                if cls is C.KFACLinearOperator:
                    # damping relative to a bound on |K|: the eigenvalues of a float32 factor of order n carry errors
                    # of ~n eps |factor| (any LAPACK-quality solver), so (K + d I)^-1 needs d well above
                    # n eps |K| to be determined at all -- 1e-3 |K| here
                    bound = 0.0
                    for block in K64[1]:
                        b = 1.0
                        for f in block:
                            b *= float(f.abs().max()) * f.shape[0]
                        bound = max(bound, b)
                    damp = 1e-3 * bound
                    if not damp > 0:
                        continue
                    for mode in ({"use_exact_damping": True}, {"use_heuristic_damping": True}):
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            try:
                                i64 = K64.inverse(damping=damp, **mode) @ V
                            except RuntimeError as err64:   # e.g. a dead layer: zero factor, Cholesky must fail
                                try:
                                    K32.inverse(damping=damp, **mode) @ V.float().to(dev)
                                    fails.append(f"{what}: inverse({mode}) raised on the CPU ({err64}) but not on the GPU")
                                except RuntimeError:
                                    pass   # same error convention on both devices
                                continue
                            i32 = K32.inverse(damping=damp, **mode) @ V.float().to(dev)
                        e = rel(i32, i64)
                        worst = max(worst, e)
                        if not e < 5e-3:
                            fails.append(f"{what}: inverse({mode}) @ V err {e:.1e}")
        except Exception as ex:  # noqa: BLE001
            fails.append(f"{what}: exception {type(ex).__name__}: {ex}")
    return worst, fails


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    worst, fails = run(seed, ncase)
    for f in fails:
        print(f)
    print(f"done: {ncase} cases, worst rel err {worst:.2e}, {len(fails)} failures")
