"""Randomised GPU fp32 vs CPU float64 comparison of the curvature operators of this package on random small nets
(the generator of tools/fuzz_kfac.py): GGN, EF, Hessian, GGN diagonal, Jacobian / transposed Jacobian, KFOC,
CG inverse of the damped GGN -- A @ V on both devices.    python tools/fuzz_ops.py [seed] [cases]"""
import os, sys, copy, warnings
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch import nn
import curvlinops_amd as C
from fuzz_kfac import make_model, rel


def run(seed, ncase):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst, fails = 0.0, []
    for case in range(ncase):
        torch.manual_seed(7000 * seed + case)
        model64, shape, out = make_model(rng)
        model64 = model64.double()
        model32 = copy.deepcopy(model64).float().to(dev)
        lossname = str(rng.choice(["mse", "ce", "bce"]))
        red = str(rng.choice(["mean", "sum"]))
        loss = {"mse": nn.MSELoss, "ce": nn.CrossEntropyLoss, "bce": nn.BCEWithLogitsLoss}[lossname](reduction=red)
        scale = 10.0 ** rng.uniform(-1, 1)
        data64 = []
        for _ in range(int(rng.integers(1, 3))):
            n = int(rng.integers(2, 9))
            X = torch.rand(n, *shape, dtype=torch.float64) * scale
            y = (torch.randint(0, out, (n,)) if lossname == "ce" else
                 torch.randint(0, 2, (n, out)).double() if lossname == "bce" else torch.rand(n, out, dtype=torch.float64))
            data64.append((X, y))
        data32 = [(X.float().to(dev), y.to(dev) if y.dtype == torch.int64 else y.float().to(dev)) for X, y in data64]
        p64, p32 = dict(model64.named_parameters()), dict(model32.named_parameters())
        what = f"case {case}: {[type(m).__name__ for m in model64]} shape {shape} loss {lossname}/{red} scale {scale:.1e}"
        ops = {
            "GGN": lambda m, l, p, d: C.GGNLinearOperator(m, l, p, d, check_deterministic=False),
            "EF": lambda m, l, p, d: C.EFLinearOperator(m, l, p, d, check_deterministic=False),
            "Hessian": lambda m, l, p, d: C.HessianLinearOperator(m, l, p, d, check_deterministic=False),
            "GGNDiagonal": lambda m, l, p, d: C.GGNDiagonalLinearOperator(m, l, p, d, check_deterministic=False),
            "Jacobian": lambda m, l, p, d: C.JacobianLinearOperator(m, p, d, check_deterministic=False),
            "TransposedJacobian": lambda m, l, p, d: C.TransposedJacobianLinearOperator(m, p, d, check_deterministic=False),
            "KFOC(type-2)": lambda m, l, p, d: C.KFOCLinearOperator(m, l, p, d, check_deterministic=False, fisher_type="type-2",
                                                                   separate_weight_and_bias=sep),
        }
        sep = bool(rng.random() < 0.5)
        if len(data64) > 1:   # KFOC: one mini-batch (the optimal rank-one factors of a sum are not the sum of the factors)
            ops = {k: v for k, v in ops.items() if not k.startswith("KFOC")}
        for name, make in ops.items():
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    A64 = make(model64, loss, p64, data64)
                    A32 = make(model32, loss, p32, data32)
                    V = torch.rand(A64.shape[1], 3, dtype=torch.float64) - 0.5
                    ref = A64 @ V
                    got = A32 @ V.float().to(dev)
                if float(ref.abs().max()) == 0.0:
                    e = float(got.abs().max())
                else:
                    e = rel(got, ref)
                worst = max(worst, e)
                if not e < (2e-2 if name.startswith("KFOC") else 1e-3):
                    fails.append(f"{what}: {name} @ V err {e:.1e}")
            except Exception as ex:  # noqa: BLE001
                fails.append(f"{what}: {name} exception {type(ex).__name__}: {str(ex)[:160]}")
    return worst, fails


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    worst, fails = run(seed, ncase)
    for f in fails:
        print(f)
    print(f"done: {ncase} cases, worst rel err {worst:.2e}, {len(fails)} failures")
