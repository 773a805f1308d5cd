"""Randomised comparison of the native MLP kernels (all batch-size regimes, vector and K-column
products, GGN and EF, aligned and unaligned widths) against the torch.func path on the same device.

    python tools/fuzz_native.py [seed] [cases]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch import nn

import curvlinops_amd as C

WIDE = float(os.environ.get("CLO_FUZZ_WIDE", "0.0"))  # fraction of cases with layers up to 1600 wide
ACTS = [nn.ReLU, nn.Tanh, nn.Sigmoid, None]
LOSSES = {"mse": nn.MSELoss, "ce": nn.CrossEntropyLoss, "bce": nn.BCEWithLogitsLoss}


def _one_case(case: int, rng, dev, failures: list) -> float:
    L = int(rng.integers(1, 5))
    align = rng.random() < 0.5
    wide = rng.random() < WIDE
    if wide:  # wide layers: several K ranges / split-K slabs per kernel
        dims = [int(rng.integers(4, 400)) * 4 for _ in range(L + 1)]
    else:
        dims = [int(rng.integers(1, 40)) * 4 if align else int(rng.integers(2, 150)) for _ in range(L + 1)]
    if rng.random() < 0.5:
        dims[-1] = int(rng.integers(1, 17))
    bias = bool(rng.random() < 0.8)
    layers = []
    for l in range(L):
        layers.append(nn.Linear(dims[l], dims[l + 1], bias=bias))
        act = ACTS[int(rng.integers(0, 4))] if l < L - 1 else None
        if act is not None:
            layers.append(act())
    torch.manual_seed(case)
    model = nn.Sequential(*layers).to(dev)
    params = dict(model.named_parameters())
    lossname = ["mse", "ce", "bce"][int(rng.integers(0, 3))]
    red = ["mean", "sum"][int(rng.integers(0, 2))]
    loss = LOSSES[lossname](reduction=red)
    data = []
    for _ in range(int(rng.integers(1, 4))):
        N = int(rng.choice([1, 3, 8, 9, 13, 16, 17, 24, 31, 32, 33, 64, 70]))
        X = torch.rand(N, dims[0], device=dev) - 0.5
        y = torch.randint(0, dims[-1], (N,), device=dev) if lossname == "ce" else torch.rand(N, dims[-1], device=dev)
        data.append((X, y))
    worst = 0.0
    classes = [C.GGNLinearOperator, C.EFLinearOperator, C.HessianLinearOperator]
    for cls in classes:
        nat = cls(model, loss, params, data, check_deterministic=False)
        assert nat.uses_native_kernels, (dims, lossname)
        ref = cls(model, loss, params, data, check_deterministic=False)
        ref._native = None
        D = nat.shape[1]
        for K in (1, int(rng.choice([3, 8, 12]))):
            V = torch.rand(D, K, device=dev) - 0.5
            what = (f"case {case}: {cls.__name__} dims={dims} bias={bias} loss={lossname}/{red} "
                    f"Ns={[x.shape[0] for x, _ in data]} K={K}")
            try:
                a = (nat @ V[:, 0].contiguous()).unsqueeze(1) if K == 1 else nat @ V
            except Exception as e:  # noqa: BLE001
                failures.append(f"exception in {what}: {e}")
                continue
            b = ref @ V
            err = float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
            worst = max(worst, err)
            if not err < 5e-4:
                failures.append(f"mismatch in {what}: err={err:.3e}")
    return worst


def run(seed: int, ncase: int):
    """Returns (worst relative error, list of failure descriptions)."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    worst, failures = 0.0, []
    for case in range(ncase):
        worst = max(worst, _one_case(case, rng, dev, failures))
    return worst, failures


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    worst, failures = run(seed, ncase)
    for f in failures:
        print(f)
    print(f"done: {ncase} cases, worst rel err {worst:.2e}, {len(failures)} failures")
