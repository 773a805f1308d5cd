"""Jacobian / transposed-Jacobian operators (reference jacobian.py:14-358): the NumPy oracle and
the operators (autograd path on CPU in float64, native kernels on the GPU in float32) against golden
vectors produced by the reference (oracle/make_golden.py jacobian)."""

import numpy as np
import pytest
import torch

import curvlinops_amd as C
from conftest import load_golden
from helpers import build_mlp, load_into, rel_err
from oracle import mlp_numpy as O

GOLD = load_golden("jacobian")
CASES = sorted(GOLD)


def _case(rec):
    dims = [int(d) for d in rec["dims"]]
    acts = [str(a) for a in rec["acts"]]
    bias = [bool(b) for b in rec["bias"]]
    lin = sorted({int(k.split(":")[1].split(".")[0]) for k in rec if k.startswith("param:")})
    Ws = [rec[f"param:{i}.weight"] for i in lin]
    bs = [rec.get(f"param:{i}.bias") for i in lin]
    Xs = [rec[f"X{i}"] for i in range(int(rec["num_batches"]))]
    return dims, acts, bias, Ws, bs, Xs


@pytest.mark.parametrize("case", CASES)
def test_oracle_jacobian_matches_reference(case):
    rec = GOLD[case]
    dims, acts, bias, Ws, bs, Xs = _case(rec)
    shapes = [W.shape for W in Ws]
    C_out = dims[-1]
    for k in range(rec["V"].shape[1]):
        vWs, vbs = O.unflatten_params(rec["V"][:, k], shapes, bias)
        got = np.concatenate([O.jacobian_matvec_batch(Ws, bs, acts, X, vWs, vbs) for X in Xs]).reshape(-1)
        assert np.abs(got - rec["J_V"][:, k]).max() <= 1e-10 * np.abs(rec["J_V"][:, k]).max() + 1e-14
    for k in range(rec["U"].shape[1]):
        U = rec["U"][:, k].reshape(-1, C_out)
        accW, accb, row = None, None, 0
        for X in Xs:
            gW, gb = O.jacobian_t_matvec_batch(Ws, bs, acts, X, U[row:row + X.shape[0]])
            row += X.shape[0]
            accW = gW if accW is None else [a + g for a, g in zip(accW, gW)]
            accb = gb if accb is None else [None if a is None else a + g for a, g in zip(accb, gb)]
        got = O.flatten_params(accW, accb)
        assert np.abs(got - rec["JT_U"][:, k]).max() <= 1e-10 * np.abs(rec["JT_U"][:, k]).max() + 1e-14


def _operators(rec, dtype, device, **kw):
    dims, acts, bias, *_ = _case(rec)
    model = build_mlp(dims, acts, bias)
    params = load_into(model, rec, dtype, device)
    data = [(torch.as_tensor(rec[f"X{i}"], dtype=dtype, device=device),
             torch.as_tensor(rec[f"y{i}"], device=device)) for i in range(int(rec["num_batches"]))]
    return (C.JacobianLinearOperator(model, params, data, **kw),
            C.TransposedJacobianLinearOperator(model, params, data, **kw))


@pytest.mark.parametrize("case", CASES)
def test_jacobian_operators_cpu(case):
    rec = GOLD[case]
    J, JT = _operators(rec, torch.float64, "cpu")
    V, U = torch.as_tensor(rec["V"]), torch.as_tensor(rec["U"])
    N_C = rec["J_V"].shape[0]
    assert J.shape == (N_C, V.shape[0]) and JT.shape == (V.shape[0], N_C)
    assert rel_err(J @ V, rec["J_V"]) < 1e-10 and rel_err(J @ V[:, 0], rec["J_v"]) < 1e-10
    assert rel_err(JT @ U, rec["JT_U"]) < 1e-10 and rel_err(JT @ U[:, 0], rec["JT_u"]) < 1e-10
    assert rel_err(J.adjoint() @ U, rec["Jadj_U"]) < 1e-10
    assert rel_err(JT.adjoint() @ V, rec["J_V"]) < 1e-10
    assert rel_err(U.T @ J, rec["JT_U"].T) < 1e-10          # left multiplication
    # tensor-list output of J has the reference's shape [(N, *out)]
    (out,) = J @ [c.reshape(*p.shape, -1) for c, p in zip(V.split([p.numel() for p in J._params.values()]),
                                                          J._params.values())]
    assert out.shape == (N_C // int(rec["dims"][-1]), int(rec["dims"][-1]), V.shape[1])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_jacobian_operators_gpu(case):
    rec = GOLD[case]
    dev = torch.device("cuda:0")
    J, JT = _operators(rec, torch.float32, dev, check_deterministic=False)
    assert J.uses_native_kernels and JT.uses_native_kernels
    V = torch.as_tensor(rec["V"], dtype=torch.float32, device=dev)
    U = torch.as_tensor(rec["U"], dtype=torch.float32, device=dev)
    assert rel_err(J @ V, rec["J_V"]) < 1e-4 and rel_err(J @ V[:, 0].contiguous(), rec["J_v"]) < 1e-4
    assert rel_err(JT @ U, rec["JT_U"]) < 1e-4 and rel_err(JT @ U[:, 0].contiguous(), rec["JT_u"]) < 1e-4
    assert rel_err(J.adjoint() @ U, rec["Jadj_U"]) < 1e-4


@pytest.mark.gpu
def test_jacobian_c2_size_gpu():
    """Full-size net (D = 10 010 122): J^T J v == G v up to the loss curvature (MSE: 2c I), adjointness."""
    from torch import nn

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(1024, 2688), nn.ReLU(), nn.Linear(2688, 2688), nn.ReLU(), nn.Linear(2688, 10)).to(dev)
    params = dict(model.named_parameters())
    data = [(torch.rand(8, 1024, device=dev), torch.rand(8, 10, device=dev)),
            (torch.rand(40, 1024, device=dev), torch.rand(40, 10, device=dev))]
    J = C.JacobianLinearOperator(model, params, data, check_deterministic=False)
    JT = J.adjoint()
    assert J.uses_native_kernels and J.shape == (480, 10_010_122)
    v, u = torch.rand(J.shape[1], device=dev) - 0.5, torch.rand(J.shape[0], device=dev) - 0.5
    Jv, JTu = J @ v, JT @ u
    lhs, rhs = torch.dot(u.double(), Jv.double()), torch.dot(JTu.double(), v.double())
    assert abs(lhs - rhs) / abs(lhs) < 1e-4
    G = C.GGNLinearOperator(model, nn.MSELoss(reduction="sum"), params, data, check_deterministic=False)
    assert rel_err(2.0 * (JT @ Jv), (G @ v).double().cpu().numpy()) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [[7, 13, 5, 3], [33, 50, 21, 6], [10, 130, 77, 4]])
def test_jacobian_odd_widths_native_gpu(dims):
    """Layer widths that are not multiples of 4 stay on the native kernels (plain GEMM products instead of the
    fused tangent-forward launch) and agree with the torch.func path on the same device."""
    from torch import nn

    import curvlinops_amd as C

    dev = torch.device("cuda:0")
    torch.manual_seed(sum(dims))
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(nn.Tanh())
    model = nn.Sequential(*layers).to(dev)
    params = dict(model.named_parameters())
    data = [(torch.rand(n, dims[0], device=dev), torch.rand(n, dims[-1], device=dev)) for n in (9, 40)]
    J = C.JacobianLinearOperator(model, params, data)
    JT = C.TransposedJacobianLinearOperator(model, params, data)
    assert J.uses_native_kernels and JT.uses_native_kernels
    J_ref = C.JacobianLinearOperator(model, params, data)
    JT_ref = C.TransposedJacobianLinearOperator(model, params, data)
    J_ref._native = JT_ref._native = None
    V = torch.rand(J.shape[1], 5, device=dev)
    U = torch.rand(J.shape[0], 3, device=dev)
    assert rel_err(J @ V, (J_ref @ V).cpu().numpy()) < 1e-5
    assert rel_err(JT @ U, (JT_ref @ U).cpu().numpy()) < 1e-5
