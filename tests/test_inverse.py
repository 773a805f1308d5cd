"""Matrix-free inverses (reference inverse.py): CG, Neumann, LSMR against dense solves; the
damped-GGN solve with a KFAC preconditioner on the GPU (reference test pattern
test/test_inverse.py:97-166)."""

import pytest
import torch
from torch import nn

import curvlinops_amd as C
from helpers import rel_err


def _spd(n, seed, dtype=torch.float64, cond=50.0):
    g = torch.Generator().manual_seed(seed)
    Q = torch.linalg.qr(torch.rand(n, n, generator=g, dtype=dtype))[0]
    lam = torch.logspace(0, -torch.log10(torch.tensor(cond)).item(), n, dtype=dtype)
    return (Q * lam) @ Q.T


def test_diagonal_operator_and_damping():
    d = [torch.rand(3, 2, dtype=torch.float64) + 0.5, torch.rand(4, dtype=torch.float64) + 0.5]
    Dg = C.DiagonalLinearOperator(d)
    x = torch.rand(10, 3, dtype=torch.float64)
    flat = torch.cat([t.flatten() for t in d])
    assert torch.allclose(Dg @ x, flat[:, None] * x)
    assert torch.allclose(Dg.inverse(0.1) @ x, x / (flat[:, None] + 0.1))
    assert isinstance(Dg + Dg, C.DiagonalLinearOperator) and torch.allclose((Dg + Dg) @ x, 2 * flat[:, None] * x)
    assert torch.allclose((Dg @ Dg) @ x, flat[:, None] ** 2 * x) and torch.allclose((3 * Dg) @ x, 3 * flat[:, None] * x)
    A = C.KroneckerProductLinearOperator(_spd(10, 0))
    damped = A + C.DiagonalLinearOperator.identity_like(A, 0.3)
    assert torch.allclose(damped @ x, A @ x + 0.3 * x)


def test_cg_inverse_matches_dense_solve():
    A = _spd(40, 1)
    op = C.KroneckerProductLinearOperator(A)
    B = torch.rand(40, 3, dtype=torch.float64)
    ref = torch.linalg.solve(A, B)
    inv = C.CGInverseLinearOperator(op, tolerance=1e-12, max_iter=200)
    assert rel_err(inv @ B, ref.numpy()) < 1e-8
    assert rel_err(inv @ B[:, 0], ref[:, 0].numpy()) < 1e-8
    assert rel_err(inv.adjoint() @ B, ref.numpy()) < 1e-8
    # a good preconditioner reaches the tolerance in far fewer iterations
    calls = {"n": 0}
    def counted(X):
        calls["n"] += 1
        return op @ X
    C.inverse.conjugate_gradients(counted, B, tolerance=1e-10, max_iter=200)
    plain = calls["n"]
    calls["n"] = 0
    Minv = C.KroneckerProductLinearOperator(torch.linalg.inv(A + 1e-3 * torch.eye(40, dtype=torch.float64)))
    X = C.inverse.conjugate_gradients(counted, B, tolerance=1e-10, max_iter=200, preconditioner=Minv.__matmul__)
    assert calls["n"] < plain / 2 and rel_err(X, ref.numpy()) < 1e-7
    with pytest.raises(TypeError):
        C.CGInverseLinearOperator(op, not_a_cg_option=1)
    with pytest.raises(ValueError):
        C.CGInverseLinearOperator(C.KroneckerProductLinearOperator(torch.rand(3, 4, dtype=torch.float64)))


def test_neumann_and_lsmr_inverse():
    A = _spd(12, 2, cond=4.0)
    op = C.KroneckerProductLinearOperator(A)
    B = torch.rand(12, 2, dtype=torch.float64)
    ref = torch.linalg.solve(A, B)
    assert rel_err(C.NeumannInverseLinearOperator(op, num_terms=400, scale=0.9) @ B, ref.numpy()) < 1e-6
    P = C.DiagonalLinearOperator([1.0 / A.diag()])
    pre = C.NeumannInverseLinearOperator(op, num_terms=300, scale=0.9, preconditioner=P.__matmul__)
    assert rel_err(pre @ B, ref.numpy()) < 1e-6 and rel_err(pre.adjoint() @ B, ref.numpy()) < 1e-6
    with pytest.raises(ValueError):  # divergent series -> NaN check
        C.NeumannInverseLinearOperator(C.KroneckerProductLinearOperator(1e3 * A), num_terms=400, scale=10.0) @ B
    assert rel_err(C.LSMRInverseLinearOperator(op, atol=1e-12, btol=1e-12) @ B, ref.numpy()) < 1e-6


@pytest.mark.gpu
def test_damped_ggn_solve_with_kfac_preconditioner_gpu():
    """(G + delta I)^-1 b on the native matvec, CG preconditioned by the damped KFAC inverse; checked
    against a dense float64 solve of the materialised operator."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(12, 16), nn.Tanh(), nn.Linear(16, 4)).to(dev)
    params = dict(model.named_parameters())
    data = [(torch.rand(8, 12, device=dev), torch.rand(8, 4, device=dev)) for _ in range(3)]
    G = C.GGNLinearOperator(model, nn.MSELoss(), params, data, check_deterministic=False)
    assert G.uses_native_kernels
    D, delta = G.shape[1], 1e-2
    damped = G + C.DiagonalLinearOperator.identity_like(G, delta)
    K = C.KFACLinearOperator(model, nn.MSELoss(), params, data, fisher_type="type-2", check_deterministic=False)
    Kinv = K.inverse(damping=delta)
    B = torch.rand(D, 2, device=dev)
    dense = (G @ torch.eye(D, device=dev)).double().cpu() + delta * torch.eye(D, dtype=torch.float64)
    ref = torch.linalg.solve(dense, B.double().cpu())
    plain = C.CGInverseLinearOperator(damped, tolerance=1e-6, max_iter=500)
    pre = C.CGInverseLinearOperator(damped, tolerance=1e-6, max_iter=500, preconditioner=Kinv.__matmul__)
    assert rel_err(plain @ B, ref.numpy()) < 1e-3
    assert rel_err(pre @ B, ref.numpy()) < 1e-3
    # single right-hand side: the fused device-resident CG kernels (clo_cg_update_f32 / _direction_f32)
    b = B[:, 0].contiguous()
    assert rel_err(plain @ b, ref[:, 0].numpy()) < 1e-3
    assert rel_err(pre @ b, ref[:, 0].numpy()) < 1e-3
    x0 = (ref[:, 0] + 0.01 * torch.rand(D, dtype=torch.float64)).float().to(dev)
    warm = C.CGInverseLinearOperator(damped, tolerance=1e-6, max_iter=500, initial_guess=x0.unsqueeze(1))
    assert rel_err(warm @ b, ref[:, 0].numpy()) < 1e-3
