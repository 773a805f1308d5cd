"""Golden vectors for the network families of BASELINE configs C4 / C5 at toy size, from the
REFERENCE (run through ``oracle/make_golden.py nets``; see there for the stub packages).

TEST INFRASTRUCTURE ONLY.  Stores inputs (parameters incl. BatchNorm statistics, data, vectors,
injected probes) and the reference's outputs.  The models are the package's own benchmark
definitions (``benchmarks/models.py``) instantiated small, so the very same module classes are
exercised at full size by ``bench.py`` / the GPU property tests:

* ``resnet_toy``  -- ResNet-style net: stem conv + BN(eval) + two BasicBlocks, the second with stride 2
  and a 1x1 down-sampling branch, global average pool, Linear head; Conv2d/Linear parameters only
  (BatchNorm excluded, as in the reference's KFAC benchmark, ``benchmark_execute.py:172-183``):
  KFAC and EKFAC, empirical + type-2 Fisher, separate and joint weight/bias.
* ``encoder_toy`` -- 2-layer pre-LN encoder, d = 32, 4 heads, ffn 64, sequences of 6, mean pool + Linear:
  ``EFLinearOperator`` products over ALL parameters, ``hutchpp_trace`` with injected probes, KFAC
  (expand) on its Linear layers.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
from torch import nn

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

torch.set_default_dtype(torch.float64)


def resnet_toy():
    from benchmarks.models import ResNetToy

    return ResNetToy()


def encoder_toy():
    from benchmarks.models import encoder_toy as make

    return make()


def _state(model) -> dict[str, np.ndarray]:
    """Every parameter AND buffer (BatchNorm running statistics) by state-dict name."""
    return {f"state:{k}": v.detach().numpy() for k, v in model.state_dict().items()}


def gen_nets(curvlinops, OUT):
    from benchmarks.models import kfac_params

    out = {}
    # ------------------------------------------------------------------ ResNet-style toy (C4 family)
    gen = torch.Generator().manual_seed(4100)
    torch.manual_seed(4100)
    model = resnet_toy()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * torch.rand(p.shape, generator=gen))
        for m in model.modules():  # non-trivial BatchNorm statistics and affine parameters
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(0.2 * torch.rand(m.running_mean.shape, generator=gen) - 0.1)
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=gen))
                m.weight.copy_(0.8 + 0.4 * torch.rand(m.weight.shape, generator=gen))
                m.bias.copy_(0.2 * torch.rand(m.bias.shape, generator=gen) - 0.1)
    model.eval()
    params = kfac_params(model)
    data = [(torch.rand(B, 3, 8, 8, generator=gen), torch.randint(0, 5, (B,), generator=gen)) for B in (5, 3)]
    D = sum(p.numel() for p in params.values())
    V = torch.rand(D, 2, generator=gen)
    rec = {"V": V.numpy(), "num_batches": np.array(len(data)), **_state(model)}
    for i, (X, y) in enumerate(data):
        rec[f"X{i}"], rec[f"y{i}"] = X.numpy(), y.numpy()
    lf = nn.CrossEntropyLoss()
    for fisher in ("empirical", "type-2"):
        for sep in (True, False):
            tag = f"{fisher}|{'sep' if sep else 'joint'}"
            K = curvlinops.KFACLinearOperator(model, lf, params, data, fisher_type=fisher,
                                              separate_weight_and_bias=sep, check_deterministic=False)
            rec[f"kfac|{tag}/KV"] = (K @ V).detach().numpy()
            for b, block in enumerate(K[1]):
                for f, fac in enumerate(block):
                    rec[f"kfac|{tag}/block{b}_factor{f}"] = fac.detach().numpy()
            rec[f"kfac|{tag}/trace"] = K.trace().numpy()
            rec[f"kfac|{tag}/inv_plain"] = (K.inverse(damping=1e-2) @ V).detach().numpy()
            rec[f"kfac|{tag}/inv_heur"] = (K.inverse(damping=1e-2, use_heuristic_damping=True, min_damping=1e-4)
                                            @ V).detach().numpy()
            rec[f"kfac|{tag}/inv_exact"] = (K.inverse(damping=1e-2, use_exact_damping=True) @ V).detach().numpy()
            E = curvlinops.EKFACLinearOperator(model, lf, params, data, fisher_type=fisher,
                                               separate_weight_and_bias=sep, check_deterministic=False)
            rec[f"ekfac|{tag}/EV"] = (E @ V).detach().numpy()
            rec[f"ekfac|{tag}/invEV"] = (E.inverse(damping=1e-2) @ V).detach().numpy()
            rec[f"ekfac|{tag}/trace"] = E.trace().numpy()
    # the exact GGN / EF of the same net (autograd path of the operators at C4's model family)
    G = curvlinops.GGNLinearOperator(model, lf, params, data, check_deterministic=False)
    rec["ggn/GV"] = (G @ V).detach().numpy()
    Fm = curvlinops.EFLinearOperator(model, lf, params, data, check_deterministic=False)
    rec["ef/FV"] = (Fm @ V).detach().numpy()
    for k, val in rec.items():
        out[f"resnet_toy/{k}"] = val

    # ------------------------------------------------------------------ encoder toy (C5 family)
    gen = torch.Generator().manual_seed(5100)
    torch.manual_seed(5100)
    model = encoder_toy()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * torch.rand(p.shape, generator=gen))
    model.eval()
    params = dict(model.named_parameters())
    data = [(torch.rand(B, 6, 32, generator=gen), torch.randint(0, 5, (B,), generator=gen)) for B in (4, 3)]
    D = sum(p.numel() for p in params.values())
    V = torch.rand(D, 3, generator=gen)
    rec = {"V": V.numpy(), "num_batches": np.array(len(data)), **_state(model)}
    for i, (X, y) in enumerate(data):
        rec[f"X{i}"], rec[f"y{i}"] = X.numpy(), y.numpy()
    lf = nn.CrossEntropyLoss()
    EF = curvlinops.EFLinearOperator(model, lf, params, data, check_deterministic=False)
    rec["ef/FV"] = (EF @ V).detach().numpy()
    rec["ef/Fv"] = (EF @ V[:, 0]).detach().numpy()
    GG = curvlinops.GGNLinearOperator(model, lf, params, data, check_deterministic=False)
    rec["ggn/GV"] = (GG @ V).detach().numpy()
    # Hutch++ with replayed probes (num_matvecs = 12 -> 4 + 4 probes)
    import curvlinops.trace.meyer2020hutch as M

    pool = torch.randint(0, 2, (D, 8), generator=gen).double() * 2 - 1
    state = {"i": 0}

    def replay(dim, distribution, device, dtype):
        v = pool[:, state["i"]].clone()
        state["i"] += 1
        return v

    orig = M.random_vector
    M.random_vector = replay
    try:
        rec["ef/hutchpp"] = M.hutchpp_trace(EF, 12, "rademacher").numpy()
    finally:
        M.random_vector = orig
    rec["ef/pool"] = pool.numpy()
    # KFAC (expand: the sequence axis is a weight-sharing axis) on the Linear layers
    from benchmarks.models import kfac_params as kp

    lin = kp(model)
    Dl = sum(p.numel() for p in lin.values())
    Vl = torch.rand(Dl, 2, generator=gen)
    rec["Vlin"] = Vl.numpy()
    for fisher in ("empirical", "type-2"):
        K = curvlinops.KFACLinearOperator(model, lf, lin, data, fisher_type=fisher, separate_weight_and_bias=False,
                                          check_deterministic=False)
        rec[f"kfac|{fisher}|joint/KV"] = (K @ Vl).detach().numpy()
        rec[f"kfac|{fisher}|joint/inv_plain"] = (K.inverse(damping=1e-2) @ Vl).detach().numpy()
        for b, block in enumerate(K[1]):
            for f, fac in enumerate(block):
                rec[f"kfac|{fisher}|joint/block{b}_factor{f}"] = fac.detach().numpy()
        E = curvlinops.EKFACLinearOperator(model, lf, lin, data, fisher_type=fisher, separate_weight_and_bias=False,
                                           check_deterministic=False)
        rec[f"ekfac|{fisher}|joint/EV"] = (E @ Vl).detach().numpy()
        rec[f"ekfac|{fisher}|joint/invEV"] = (E.inverse(damping=1e-2) @ Vl).detach().numpy()
    for k, val in rec.items():
        out[f"encoder_toy/{k}"] = val
    np.savez_compressed(OUT / "nets.npz", **out)
    print("nets.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def gen_kfac_mc(curvlinops, OUT):
    """``fisher_type="mc"`` on the two convolutional benchmark families (LeNet-5 = BASELINE config C3 at B = 6 + 4;
    the ResNet toy of `gen_nets`) with the reference's SAMPLED backprop vectors captured: torch's CPU and GPU random
    streams differ, so the test replays the captured ``[M, B, C]`` tensors and compares the factors, the
    product and the damped inverse of the exact MC code path -- V = mc_samples vectors per datum scaled by 1/sqrt(M),
    the mean-reduction correction -- with the reference's own numbers instead of "equal in expectation"."""
    from benchmarks.models import kfac_params, lenet5
    from curvlinops.computers import _base

    captured = []
    original = _base._BaseKFACComputer._set_up_grad_outputs_computer

    def capturing(loss_func, fisher_type, mc_samples):
        fn = original(loss_func, fisher_type, mc_samples)

        def wrapped(output, y, generator):
            g = fn(output, y, generator)
            captured.append(g.detach().clone())
            return g

        return wrapped

    out = {}
    cases = []
    gen = torch.Generator().manual_seed(7100)
    torch.manual_seed(7100)
    net = lenet5()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.rand(p.shape, generator=gen))
    data = [(torch.rand(B, 1, 32, 32, generator=gen), torch.randint(0, 10, (B,), generator=gen)) for B in (6, 4)]
    cases.append(("lenet5", net, dict(net.named_parameters()), data))
    gen = torch.Generator().manual_seed(7200)
    torch.manual_seed(7200)
    net = resnet_toy()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.rand(p.shape, generator=gen))
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(0.2 * torch.rand(m.running_mean.shape, generator=gen) - 0.1)
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=gen))
    net.eval()
    data = [(torch.rand(B, 3, 8, 8, generator=gen), torch.randint(0, 5, (B,), generator=gen)) for B in (5, 3)]
    cases.append(("resnet_toy", net, kfac_params(net), data))

    _base._BaseKFACComputer._set_up_grad_outputs_computer = staticmethod(capturing)
    try:
        for name, model, params, data in cases:
            D = sum(p.numel() for p in params.values())
            V = torch.rand(D, 2, generator=gen)
            rec = {"V": V.numpy(), "num_batches": np.array(len(data)), **_state(model)}
            for i, (X, y) in enumerate(data):
                rec[f"X{i}"], rec[f"y{i}"] = X.numpy(), y.numpy()
            lf = nn.CrossEntropyLoss()
            for M in (1, 3):
                # (LeNet-5's factors are large for a fixture: joint weight + bias only, as BASELINE config C3 runs it)
                for sep in ((False,) if name == "lenet5" else (True, False)):
                    tag = f"mc{M}|{'sep' if sep else 'joint'}"
                    captured.clear()
                    K = curvlinops.KFACLinearOperator(model, lf, params, data, fisher_type="mc", mc_samples=M,
                                                      separate_weight_and_bias=sep, check_deterministic=False)
                    assert len(captured) == len(data), (len(captured), len(data))
                    for i, g in enumerate(captured):
                        rec[f"{tag}/grad_outputs{i}"] = g.numpy()      # [M, B, C], before the 1/B of the mean
                    rec[f"{tag}/KV"] = (K @ V).detach().numpy()
                    for b, block in enumerate(K[1]):
                        for f, fac in enumerate(block):
                            rec[f"{tag}/block{b}_factor{f}"] = fac.detach().numpy()
                    rec[f"{tag}/inv_plain"] = (K.inverse(damping=1e-2) @ V).detach().numpy()
            for k, val in rec.items():   # large arrays in float32 (the GPU comparison is at 1e-4)
                out[f"{name}/{k}"] = val.astype(np.float32) if val.dtype == np.float64 and val.size > 20000 else val
    finally:
        _base._BaseKFACComputer._set_up_grad_outputs_computer = staticmethod(original)
    np.savez_compressed(OUT / "kfac_mc.npz", **out)
    print("kfac_mc.npz:", len(out), "arrays")
