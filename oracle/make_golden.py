"""Generate golden vectors by running the REFERENCE (f-dangel/curvlinops) in this container.

TEST INFRASTRUCTURE ONLY.  Needs ``/root/reference`` and two tiny stub packages for
third-party imports that are absent here (``einconv``, ``linear_operator``):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/stubs:/root/reference \
        python -B oracle/make_golden.py

Outputs ``tests/golden/*.npz`` (inputs + reference outputs, float64).  Only data is stored --
no reference source.  The reference never travels to the GPU box; tests read the .npz files.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
from torch import nn

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
torch.set_default_dtype(torch.float64)

ACT = {"identity": None, "relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}
LOSS = {"mse": nn.MSELoss, "ce": nn.CrossEntropyLoss, "bce": nn.BCEWithLogitsLoss}


def build_mlp(dims, acts, bias):
    layers = []
    for i, act in enumerate(acts):
        layers.append(nn.Linear(dims[i], dims[i + 1], bias=bias[i]))
        if ACT[act] is not None:
            layers.append(ACT[act]())
    return nn.Sequential(*layers)


def make_data(gen, batch_sizes, d_in, C, loss):
    data = []
    for B in batch_sizes:
        X = torch.rand(B, d_in, generator=gen)
        if loss == "ce":
            y = torch.randint(0, C, (B,), generator=gen)
        elif loss == "bce":
            y = torch.randint(0, 2, (B, C), generator=gen).double()
        else:
            y = torch.rand(B, C, generator=gen)
        data.append((X, y))
    return data


MLP_CASES = [
    # name, dims, acts, bias, loss, reduction, batch sizes
    ("c1_tanh_mse_mean", [128, 256, 64, 10], ["tanh", "tanh", "identity"], [True] * 3, "mse", "mean", [64, 64]),
    ("relu_mse_mean", [12, 16, 20, 5], ["relu", "relu", "identity"], [True] * 3, "mse", "mean", [7, 3]),
    ("relu_mse_sum", [12, 16, 20, 5], ["relu", "relu", "identity"], [True, False, True], "mse", "sum", [5, 9]),
    ("sigm_ce_mean", [10, 9, 7], ["sigmoid", "identity"], [True, True], "ce", "mean", [6, 2]),
    ("tanh_ce_sum", [10, 9, 7], ["tanh", "identity"], [False, True], "ce", "sum", [4, 8]),
    ("relu_bce_mean", [11, 8, 8, 3], ["relu", "tanh", "identity"], [True] * 3, "bce", "mean", [3, 10]),
    ("sigm_out_mse_mean", [6, 5, 4], ["tanh", "sigmoid"], [True, True], "mse", "mean", [9]),
    ("odd_dims_mse_mean", [13, 17, 3], ["relu", "identity"], [True, True], "mse", "mean", [1, 8, 16]),
]


def gen_mlp(curvlinops):
    out = {}
    for idx, (name, dims, acts, bias, loss, red, bsz) in enumerate(MLP_CASES):
        gen = torch.Generator().manual_seed(1000 + idx)
        torch.manual_seed(1000 + idx)
        model = build_mlp(dims, acts, bias)
        for p in model.parameters():
            p.data += 0.01 * torch.rand(p.shape, generator=gen)
        data = make_data(gen, bsz, dims[0], dims[-1], loss)
        params = dict(model.named_parameters())
        D = sum(p.numel() for p in params.values())
        v = torch.rand(D, generator=gen)
        # the 50k-parameter C1 case keeps only the vector product (fixture size)
        V = torch.rand(D, 1 if D > 20000 else 3, generator=gen)
        loss_func = LOSS[loss](reduction=red)
        rec = {
            "dims": np.array(dims), "acts": np.array(acts), "bias": np.array(bias),
            "loss": np.array(loss), "reduction": np.array(red), "v": v.numpy(), "V": V.numpy(),
            "num_batches": np.array(len(data)),
        }
        for i, (X, y) in enumerate(data):
            rec[f"X{i}"] = X.numpy()
            rec[f"y{i}"] = y.numpy()
        for k, p in params.items():
            rec[f"param:{k}"] = p.detach().numpy()
        for opname, cls in (("ggn", curvlinops.GGNLinearOperator),
                            ("hessian", curvlinops.HessianLinearOperator),
                            ("ef", curvlinops.EFLinearOperator)):
            op = cls(model, loss_func, params, data)
            rec[f"{opname}_v"] = (op @ v).detach().numpy()
            rec[f"{opname}_V"] = (op @ V).detach().numpy()
        for k, val in rec.items():
            out[f"{name}/{k}"] = val
    np.savez_compressed(OUT / "mlp_curvature.npz", **out)
    print("mlp_curvature.npz:", len(out), "arrays")


def gen_jacobian(curvlinops):
    """Jacobian / transposed-Jacobian products (jacobian.py:14-358) on the MLP cases above (same
    seeds => same nets and data as mlp_curvature.npz; inputs are stored again for self-containment)."""
    out = {}
    for idx, (name, dims, acts, bias, loss, red, bsz) in enumerate(MLP_CASES):
        if dims[0] > 64:
            continue  # fixture size
        gen = torch.Generator().manual_seed(1000 + idx)
        torch.manual_seed(1000 + idx)
        model = build_mlp(dims, acts, bias)
        for p in model.parameters():
            p.data += 0.01 * torch.rand(p.shape, generator=gen)
        data = make_data(gen, bsz, dims[0], dims[-1], loss)
        params = dict(model.named_parameters())
        D = sum(p.numel() for p in params.values())
        g2 = torch.Generator().manual_seed(5000 + idx)
        V = torch.rand(D, 3, generator=g2)
        N = sum(bsz)
        U = torch.rand(N * dims[-1], 2, generator=g2)
        rec = {"dims": np.array(dims), "acts": np.array(acts), "bias": np.array(bias), "V": V.numpy(),
               "U": U.numpy(), "num_batches": np.array(len(data))}
        for i, (X, y) in enumerate(data):
            rec[f"X{i}"] = X.numpy()
            rec[f"y{i}"] = y.numpy()
        for k, p in params.items():
            rec[f"param:{k}"] = p.detach().numpy()
        J = curvlinops.JacobianLinearOperator(model, params, data)
        JT = curvlinops.TransposedJacobianLinearOperator(model, params, data)
        rec["J_V"] = (J @ V).detach().numpy()
        rec["J_v"] = (J @ V[:, 0]).detach().numpy()
        rec["JT_U"] = (JT @ U).detach().numpy()
        rec["JT_u"] = (JT @ U[:, 0]).detach().numpy()
        rec["Jadj_U"] = (J.adjoint() @ U).detach().numpy()
        for k, val in rec.items():
            out[f"{name}/{k}"] = val
    np.savez_compressed(OUT / "jacobian.npz", **out)
    print("jacobian.npz:", len(out), "arrays")


def gen_ggn_diagonal(curvlinops):
    """Exact GGN diagonal (ggn_diagonal.py / computers/ggn_diagonal.py) on small MLPs and a CNN."""
    out = {}
    cases = [(i, c) for i, c in enumerate(MLP_CASES) if c[1][0] <= 64]
    for idx, (name, dims, acts, bias, loss, red, bsz) in cases:
        gen = torch.Generator().manual_seed(1000 + idx)
        torch.manual_seed(1000 + idx)
        model = build_mlp(dims, acts, bias)
        for p in model.parameters():
            p.data += 0.01 * torch.rand(p.shape, generator=gen)
        data = make_data(gen, bsz, dims[0], dims[-1], loss)
        params = dict(model.named_parameters())
        op = curvlinops.GGNDiagonalLinearOperator(model, LOSS[loss](reduction=red), params, data)
        rec = {"dims": np.array(dims), "acts": np.array(acts), "bias": np.array(bias), "loss": np.array(loss),
               "reduction": np.array(red), "num_batches": np.array(len(data)), "kind": np.array("mlp"),
               "diag": torch.cat([d.flatten() for d in op._diagonal]).detach().numpy()}
        for i, (X, y) in enumerate(data):
            rec[f"X{i}"], rec[f"y{i}"] = X.numpy(), y.numpy()
        for k, p in params.items():
            rec[f"param:{k}"] = p.detach().numpy()
        for k, val in rec.items():
            out[f"{name}/{k}"] = val
    # a small CNN: conv (padding, stride) -> relu -> flatten -> linear, CE mean
    gen = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    model = nn.Sequential(nn.Conv2d(2, 3, 3, padding=1, stride=2), nn.ReLU(), nn.Flatten(start_dim=-3),
                          nn.Linear(3 * 3 * 3, 4))
    data = [(torch.rand(B, 2, 6, 6, generator=gen), torch.randint(0, 4, (B,), generator=gen)) for B in (3, 5)]
    params = dict(model.named_parameters())
    op = curvlinops.GGNDiagonalLinearOperator(model, nn.CrossEntropyLoss(), params, data)
    rec = {"kind": np.array("cnn"), "num_batches": np.array(2),
           "diag": torch.cat([d.flatten() for d in op._diagonal]).detach().numpy()}
    for i, (X, y) in enumerate(data):
        rec[f"X{i}"], rec[f"y{i}"] = X.numpy(), y.numpy()
    for k, p in params.items():
        rec[f"param:{k}"] = p.detach().numpy()
    for k, val in rec.items():
        out[f"cnn_ce_mean/{k}"] = val
    np.savez_compressed(OUT / "ggn_diagonal.npz", **out)
    print("ggn_diagonal.npz:", len(out), "arrays")


def gen_linops(curvlinops):
    """Kronecker / eigendecomposed / block-diagonal / canonical-converter known answers."""
    from curvlinops.blockdiagonal import BlockDiagonalLinearOperator
    from curvlinops.eigh import EighDecomposedLinearOperator
    from curvlinops.kfac_utils import ToCanonicalLinearOperator
    from curvlinops.kronecker import KroneckerProductLinearOperator

    gen = torch.Generator().manual_seed(7)
    out = {}
    # rectangular and square Kronecker factors
    for name, shapes in (("rect", [(5, 3), (4, 6)]), ("sq", [(6, 6), (9, 9)]), ("one", [(7, 7)]),
                         ("three", [(2, 3), (4, 2), (3, 3)])):
        fs = [torch.rand(*s, generator=gen) for s in shapes]
        if name in ("sq", "one"):
            fs = [f @ f.T + 0.1 * torch.eye(f.shape[0]) for f in fs]
        K = KroneckerProductLinearOperator(*fs)
        X = torch.rand(K.shape[1], 4, generator=gen)
        Y = torch.rand(K.shape[0], 4, generator=gen)
        for i, f in enumerate(fs):
            out[f"kron_{name}/factor{i}"] = f.numpy()
        out[f"kron_{name}/X"] = X.numpy()
        out[f"kron_{name}/KX"] = (K @ X).numpy()
        out[f"kron_{name}/Y"] = Y.numpy()
        out[f"kron_{name}/KTY"] = (K.adjoint() @ Y).numpy()
        if name in ("sq", "one"):
            out[f"kron_{name}/trace"] = K.trace().numpy()
            out[f"kron_{name}/det"] = K.det().numpy()
            out[f"kron_{name}/logdet"] = K.logdet().numpy()
            out[f"kron_{name}/fro"] = K.frobenius_norm().numpy()
            out[f"kron_{name}/inv_plain_X"] = (K.inverse(damping=1e-2) @ X).numpy()
            if len(fs) <= 2:
                out[f"kron_{name}/inv_heur_X"] = (
                    K.inverse(damping=1e-2, use_heuristic_damping=True, min_damping=1e-3) @ X
                ).numpy()
            out[f"kron_{name}/inv_exact_X"] = (K.inverse(damping=1e-2, use_exact_damping=True) @ X).numpy()
    # eigendecomposed operator with a Kronecker eigenbasis
    Q1 = torch.linalg.qr(torch.rand(4, 4, generator=gen))[0]
    Q2 = torch.linalg.qr(torch.rand(5, 5, generator=gen))[0]
    lam = torch.rand(20, generator=gen) + 0.1
    E = EighDecomposedLinearOperator(lam, KroneckerProductLinearOperator(Q1, Q2))
    X = torch.rand(20, 3, generator=gen)
    out.update({"eigh/Q1": Q1.numpy(), "eigh/Q2": Q2.numpy(), "eigh/lam": lam.numpy(), "eigh/X": X.numpy(),
                "eigh/EX": (E @ X).numpy(), "eigh/invEX": (E.inverse(damping=0.05) @ X).numpy(),
                "eigh/trace": E.trace().numpy(), "eigh/logdet": E.logdet().numpy(),
                "eigh/fro": E.frobenius_norm().numpy(), "eigh/det": E.det().numpy()})
    # block diagonal of two Kronecker blocks
    A1, A2 = torch.rand(3, 3, generator=gen), torch.rand(4, 4, generator=gen)
    B1 = torch.rand(5, 5, generator=gen)
    BD = BlockDiagonalLinearOperator([KroneckerProductLinearOperator(A1, A2), KroneckerProductLinearOperator(B1)])
    X = torch.rand(17, 2, generator=gen)
    out.update({"bd/A1": A1.numpy(), "bd/A2": A2.numpy(), "bd/B1": B1.numpy(), "bd/X": X.numpy(),
                "bd/BDX": (BD @ X).numpy(), "bd/trace": BD.trace().numpy(), "bd/fro": BD.frobenius_norm().numpy()})
    # canonical converters with shuffled parameter order (test/test_kfac_utils.py:23-34)
    shapes = {"l2.bias": torch.Size([4]), "l1.weight": torch.Size([3, 5]), "c.weight": torch.Size([2, 3, 2, 2]),
              "l2.weight": torch.Size([4, 3]), "c.bias": torch.Size([2]), "l1.bias": torch.Size([3])}
    groups = [{"W": "l1.weight", "b": "l1.bias"}, {"W": "c.weight", "b": "c.bias"}, {"W": "l2.weight"}, {"b": "l2.bias"}]
    PT = ToCanonicalLinearOperator(shapes, groups, torch.device("cpu"), torch.float64)
    D = sum(s.numel() for s in shapes.values())
    X = torch.rand(D, 2, generator=gen)
    out.update({"canon/X": X.numpy(), "canon/PTX": (PT @ X).numpy(), "canon/PPTX": (PT.adjoint() @ (PT @ X)).numpy()})
    np.savez_compressed(OUT / "linops.npz", **out)
    print("linops.npz:", len(out), "arrays")


# K-column products on shapes the native column kernels accept (layer inputs % 4 == 0, K % 4 == 0, <= 32 rows per
# mini-batch, linear last layer): `clo_mlp_ggn_matmat` / `clo_mlp_hessian_matmat` against the reference's vmap
# (_torch_base.py:946-989 over ggn.py:41-72, hessian.py:66, gradient_moments.py:48-87) with K = 8 columns.
COLUMN_CASES = [
    # name, dims, acts, bias, loss, reduction, batch sizes
    ("cols_tanh_mse_mean", [16, 24, 20, 5], ["tanh", "sigmoid", "identity"], [True] * 3, "mse", "mean", [8, 5]),
    ("cols_relu_ce_mean", [12, 16, 8, 7], ["relu", "tanh", "identity"], [True, False, True], "ce", "mean", [6, 11]),
    ("cols_sigm_bce_sum", [8, 12, 3], ["sigmoid", "identity"], [True, True], "bce", "sum", [9]),
]


def gen_columns(curvlinops):
    out = {}
    for idx, (name, dims, acts, bias, loss, red, bsz) in enumerate(COLUMN_CASES):
        gen = torch.Generator().manual_seed(4000 + idx)
        torch.manual_seed(4000 + idx)
        model = build_mlp(dims, acts, bias)
        for p in model.parameters():
            p.data += 0.01 * torch.rand(p.shape, generator=gen)
        data = make_data(gen, bsz, dims[0], dims[-1], loss)
        params = dict(model.named_parameters())
        D = sum(p.numel() for p in params.values())
        V = torch.rand(D, 8, generator=gen) - 0.5
        loss_func = LOSS[loss](reduction=red)
        rec = {
            "dims": np.array(dims), "acts": np.array(acts), "bias": np.array(bias),
            "loss": np.array(loss), "reduction": np.array(red), "V": V.numpy(),
            "num_batches": np.array(len(data)),
        }
        for i, (X, y) in enumerate(data):
            rec[f"X{i}"] = X.numpy()
            rec[f"y{i}"] = y.numpy()
        for k, p in params.items():
            rec[f"param:{k}"] = p.detach().numpy()
        for opname, cls in (("ggn", curvlinops.GGNLinearOperator),
                            ("hessian", curvlinops.HessianLinearOperator),
                            ("ef", curvlinops.EFLinearOperator)):
            rec[f"{opname}_V"] = (cls(model, loss_func, params, data) @ V).detach().numpy()
        for k, val in rec.items():
            out[f"{name}/{k}"] = val
    np.savez_compressed(OUT / "mlp_columns.npz", **out)
    print("mlp_columns.npz:", len(out), "arrays")


def main():
    if not Path("/root/reference/curvlinops").exists():
        sys.exit("reference not present: golden vectors can only be regenerated in the build container")
    import curvlinops

    OUT.mkdir(parents=True, exist_ok=True)
    which = sys.argv[1:] or ["mlp", "columns", "jacobian", "ggn_diagonal", "linops", "kfac", "trace", "trace_decay", "kfoc", "nets", "kfac_mc"]
    if "mlp" in which:
        gen_mlp(curvlinops)
    if "columns" in which:
        gen_columns(curvlinops)
    if "jacobian" in which:
        gen_jacobian(curvlinops)
    if "ggn_diagonal" in which:
        gen_ggn_diagonal(curvlinops)
    if "linops" in which:
        gen_linops(curvlinops)
    if "kfac" in which:
        from make_golden_kfac import gen_kfac

        gen_kfac(curvlinops, OUT)
    if "trace" in which:
        from make_golden_kfac import gen_trace

        gen_trace(curvlinops, OUT)
    if "trace_decay" in which:
        from make_golden_kfac import gen_trace_decay

        gen_trace_decay(curvlinops, OUT)
    if "kfoc" in which:
        from make_golden_kfac import gen_kfoc

        gen_kfoc(curvlinops, OUT)
    if "kfac_mc" in which:
        from make_golden_nets import gen_kfac_mc

        gen_kfac_mc(curvlinops, OUT)
    if "nets" in which:
        from make_golden_nets import gen_nets

        gen_nets(curvlinops, OUT)


if __name__ == "__main__":
    main()
