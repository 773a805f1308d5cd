"""KFAC / EKFAC / trace golden vectors from the REFERENCE (see make_golden.py for how to run).

TEST INFRASTRUCTURE ONLY.  Stores inputs (parameters, data, vectors, injected probes) and the
reference's outputs (Kronecker factors, products, damped-inverse products, scalar properties).
"""

from __future__ import annotations

import numpy as np
import torch
from torch import nn

torch.set_default_dtype(torch.float64)


class Unsqueeze(nn.Module):  # not used by the product; only to build reference models
    pass


def mlp(dims, act=nn.ReLU, bias=True):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1], bias=bias))
        if i < len(dims) - 2:
            layers.append(act())
    return nn.Sequential(*layers)


def cnn():
    return nn.Sequential(
        nn.Conv2d(2, 3, 3, padding=1), nn.ReLU(), nn.Conv2d(3, 4, 3, stride=2), nn.Sigmoid(),
        nn.Flatten(), nn.Linear(4 * 3 * 3, 5),
    )


KFAC_CASES = [
    # name, model factory, input shape (w/o batch), C, loss, reduction, batches, extra kfac kwargs
    ("mlp_mse_mean", lambda: mlp([7, 9, 6, 3]), (7,), 3, "mse", "mean", [12, 20], {}),
    ("mlp_ce_sum", lambda: mlp([7, 9, 6, 4], act=nn.Tanh), (7,), 4, "ce", "sum", [16, 16], {}),
    ("mlp_bce_mean_nobias", lambda: mlp([6, 8, 3], bias=False), (6,), 3, "bce", "mean", [30], {}),
    ("cnn_ce_mean", cnn, (2, 8, 8), 5, "ce", "mean", [6, 10], {}),
    ("seq_mse_mean", lambda: mlp([5, 6, 2]), (4, 5), 2, "mse", "mean", [9, 7], {}),
]
LOSS = {"mse": nn.MSELoss, "ce": nn.CrossEntropyLoss, "bce": nn.BCEWithLogitsLoss}


def _data(gen, batches, in_shape, C, loss, seq):
    out = []
    for B in batches:
        X = torch.rand(B, *in_shape, generator=gen)
        lead = (B, *in_shape[:-1]) if seq else (B,)
        if loss == "ce":
            y = torch.randint(0, C, lead, generator=gen)
        elif loss == "bce":
            y = torch.randint(0, 2, (*lead, C), generator=gen).double()
        else:
            y = torch.rand(*lead, C, generator=gen)
        out.append((X, y))
    return out


def gen_kfac(curvlinops, OUT):
    out = {}
    for idx, (name, factory, in_shape, C, loss, red, batches, _) in enumerate(KFAC_CASES):
        gen = torch.Generator().manual_seed(500 + idx)
        torch.manual_seed(500 + idx)
        model = factory()
        for p in model.parameters():
            p.data += 0.01 * torch.rand(p.shape, generator=gen)
        seq = name.startswith("seq")
        data = _data(gen, batches, in_shape, C, loss, seq)
        params = dict(model.named_parameters())
        D = sum(p.numel() for p in params.values())
        V = torch.rand(D, 2, generator=gen)
        rec = {"loss": np.array(loss), "reduction": np.array(red), "V": V.numpy(),
               "num_batches": np.array(len(data)), "in_shape": np.array(in_shape), "C": np.array(C)}
        for i, (X, y) in enumerate(data):
            rec[f"X{i}"], rec[f"y{i}"] = X.numpy(), y.numpy()
        for k, p in params.items():
            rec[f"param:{k}"] = p.detach().numpy()
        loss_func = LOSS[loss](reduction=red)
        approxes = ["expand", "reduce"] if name.startswith(("cnn", "seq")) else ["expand"]
        for fisher in ("type-2", "empirical", "forward-only"):
            for approx in approxes:
                for sep in (True, False):
                    tag = f"{fisher}|{approx}|{'sep' if sep else 'joint'}"
                    K = curvlinops.KFACLinearOperator(
                        model, loss_func, params, data, fisher_type=fisher, kfac_approx=approx,
                        separate_weight_and_bias=sep, check_deterministic=False,
                    )
                    rec[f"{tag}/KV"] = (K @ V).detach().numpy()
                    _, Kc, _ = K
                    for b, block in enumerate(Kc):
                        for f, fac in enumerate(block):
                            rec[f"{tag}/block{b}_factor{f}"] = fac.detach().numpy()
                    rec[f"{tag}/trace"] = K.trace().numpy()
                    rec[f"{tag}/fro"] = K.frobenius_norm().numpy()
                    rec[f"{tag}/inv_plain"] = (K.inverse(damping=1e-2) @ V).detach().numpy()
                    rec[f"{tag}/inv_exact"] = (K.inverse(damping=1e-2, use_exact_damping=True) @ V).detach().numpy()
                    if fisher != "forward-only" or True:
                        try:
                            rec[f"{tag}/inv_heur"] = (
                                K.inverse(damping=1e-2, use_heuristic_damping=True, min_damping=1e-4) @ V
                            ).detach().numpy()
                        except (RuntimeError, ValueError):
                            pass
        # EKFAC needs 2-d outputs; eigen-decompositions are unique only for simple spectra, so
        # the product is stored (basis-independent then), not the bases.
        if not seq:
            for fisher in ("type-2", "empirical"):
                for sep in (True, False):
                    tag = f"ekfac|{fisher}|{'sep' if sep else 'joint'}"
                    E = curvlinops.EKFACLinearOperator(
                        model, loss_func, params, data, fisher_type=fisher,
                        separate_weight_and_bias=sep, check_deterministic=False,
                    )
                    rec[f"{tag}/EV"] = (E @ V).detach().numpy()
                    rec[f"{tag}/invEV"] = (E.inverse(damping=1e-2) @ V).detach().numpy()
                    rec[f"{tag}/trace"] = E.trace().numpy()
        for k, val in rec.items():
            out[f"{name}/{k}"] = val
    np.savez_compressed(OUT / "kfac.npz", **out)
    print("kfac.npz:", len(out), "arrays")


def gen_trace(curvlinops, OUT):
    """Hutchinson / Hutch++ with INJECTED probes: the reference's sampler is replaced by one
    that replays pre-drawn probe vectors, which are stored next to the estimates."""
    import curvlinops.trace.hutchinson as H
    import curvlinops.trace.meyer2020hutch as M
    from curvlinops.examples import TensorLinearOperator

    gen = torch.Generator().manual_seed(99)
    B = torch.rand(30, 30, generator=gen)
    A = B @ B.T
    out = {"A": A.numpy()}
    for dist in ("rademacher", "normal"):
        pool = (torch.randint(0, 2, (30, 24), generator=gen).double() * 2 - 1) if dist == "rademacher" \
            else torch.randn(30, 24, generator=gen)
        state = {"i": 0}

        def replay(dim, distribution, device, dtype, pool=pool, state=state):
            v = pool[:, state["i"]].clone()
            state["i"] += 1
            return v

        orig_h, orig_m = H.random_vector, M.random_vector
        H.random_vector = replay
        M.random_vector = replay
        try:
            op = TensorLinearOperator(A)
            state["i"] = 0
            out[f"{dist}/hutch"] = H.hutchinson_trace(op, 12, dist).numpy()
            state["i"] = 0
            out[f"{dist}/hutchpp"] = M.hutchpp_trace(op, 24, dist).numpy()
        finally:
            H.random_vector, M.random_vector = orig_h, orig_m
        # diagonal and squared-Frobenius-norm estimators replay the same pool from its start
        import curvlinops.diagonal.hutchinson as DH
        import curvlinops.norm.hutchinson as NH

        orig_d, orig_n = DH.random_vector, NH.random_vector
        DH.random_vector = replay
        NH.random_vector = replay
        try:
            state["i"] = 0
            out[f"{dist}/hutch_diag"] = DH.hutchinson_diag(op, 12, dist).numpy()
            state["i"] = 0
            out[f"{dist}/hutch_fro2"] = NH.hutchinson_squared_fro(op, 12, dist).numpy()
        finally:
            DH.random_vector, NH.random_vector = orig_d, orig_n
        import curvlinops.diagonal.epperly2024xtrace as XD
        import curvlinops.trace.epperly2024xtrace as XT

        orig_xt, orig_xd = XT.random_vector, XD.random_vector
        XT.random_vector = replay
        XD.random_vector = replay
        try:
            state["i"] = 0
            out[f"{dist}/xtrace"] = XT.xtrace(op, 16, dist).numpy()
            if dist == "rademacher":
                state["i"] = 0
                out[f"{dist}/xdiag"] = XD.xdiag(op, 16).numpy()
        finally:
            XT.random_vector, XD.random_vector = orig_xt, orig_xd
        out[f"{dist}/pool"] = pool.numpy()
    np.savez_compressed(OUT / "trace.npz", **{f"t/{k}": v for k, v in out.items()})
    print("trace.npz:", len(out), "arrays")


def gen_trace_decay(curvlinops, OUT):
    """Hutch++ / XTrace with injected probes on an operator whose spectrum DECAYS over six decades inside the sketch
    (``A = U diag(lam) U^T``, 40 eigenvalues 1 ... 1e-6 in 512 dimensions, 32 probe columns per block): the regime in
    which a range basis that drops small directions differs from the reference's Householder ``Q`` (all n columns,
    ``meyer2020hutch.py:89-93``).  Stored: the factors of ``A`` (not the 512 x 512 matrix), the probes, the estimates."""
    import curvlinops.trace.epperly2024xtrace as XT
    import curvlinops.trace.meyer2020hutch as M
    from curvlinops.examples import TensorLinearOperator

    gen = torch.Generator().manual_seed(4242)
    D, R, N = 512, 40, 32
    U = torch.linalg.qr(torch.randn(D, R, generator=gen))[0]
    lam = 10.0 ** (-6.0 * torch.arange(R, dtype=torch.float64) / (R - 1))
    A = (U * lam) @ U.T
    out = {"U": U.numpy(), "lam": lam.numpy(), "trace_exact": lam.sum().numpy()}
    for dist in ("rademacher", "normal"):
        pool = (torch.randint(0, 2, (D, 2 * N), generator=gen).double() * 2 - 1) if dist == "rademacher" \
            else torch.randn(D, 2 * N, generator=gen)
        state = {"i": 0}

        def replay(dim, distribution, device, dtype, pool=pool, state=state):
            v = pool[:, state["i"]].clone()
            state["i"] += 1
            return v

        orig_m, orig_x = M.random_vector, XT.random_vector
        M.random_vector = replay
        XT.random_vector = replay
        try:
            op = TensorLinearOperator(A)
            state["i"] = 0
            out[f"{dist}/hutchpp"] = M.hutchpp_trace(op, 3 * N, dist).numpy()
            state["i"] = 0
            out[f"{dist}/xtrace"] = XT.xtrace(op, 2 * N, dist).numpy()
        finally:
            M.random_vector, XT.random_vector = orig_m, orig_x
        out[f"{dist}/pool"] = pool.numpy()
    np.savez_compressed(OUT / "trace_decay.npz", **{f"t/{k}": v for k, v in out.items()})
    print("trace_decay.npz:", len(out), "arrays")


KFOC_CASES = [
    # name, model factory, input shape, C, loss, reduction, batch
    ("mlp_mse_mean", lambda: mlp([7, 9, 6, 3]), (7,), 3, "mse", "mean", 12),
    ("mlp_ce_mean", lambda: mlp([7, 9, 6, 4], act=nn.Tanh), (7,), 4, "ce", "mean", 10),
    ("mlp_bce_sum_nobias", lambda: mlp([6, 8, 3], bias=False), (6,), 3, "bce", "sum", 9),
    ("cnn_ce_mean", cnn, (2, 8, 8), 5, "ce", "mean", 6),
]


def gen_kfoc(curvlinops, OUT):
    """KFOC (type-2: deterministic) on a single batch: the dense operator, which does not depend on
    the sign ARPACK happens to return for a singular pair, and its product with fixed vectors."""
    out = {}
    for idx, (name, factory, in_shape, C, loss, red, B) in enumerate(KFOC_CASES):
        gen = torch.Generator().manual_seed(900 + idx)
        torch.manual_seed(900 + idx)
        model = factory()
        for p in model.parameters():
            p.data += 0.01 * torch.rand(p.shape, generator=gen)
        (X, y), = _data(gen, [B], in_shape, C, loss, False)
        params = dict(model.named_parameters())
        D = sum(p.numel() for p in params.values())
        V = torch.rand(D, 2, generator=gen)
        rec = {"loss": np.array(loss), "reduction": np.array(red), "V": V.numpy(), "X": X.numpy(),
               "y": y.numpy(), "in_shape": np.array(in_shape), "C": np.array(C)}
        for k, p in params.items():
            rec[f"param:{k}"] = p.detach().numpy()
        for sep in (True, False):
            K = curvlinops.KFOCLinearOperator(model, LOSS[loss](reduction=red), params, [(X, y)],
                                              fisher_type="type-2", separate_weight_and_bias=sep)
            tag = "sep" if sep else "joint"
            rec[f"{tag}/dense"] = (K @ torch.eye(D)).detach().numpy()
            rec[f"{tag}/KV"] = (K @ V).detach().numpy()
        for k, val in rec.items():
            out[f"{name}/{k}"] = val
    np.savez_compressed(OUT / "kfoc.npz", **out)
    print("kfoc.npz:", len(out), "arrays")
