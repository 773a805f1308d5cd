"""Matrix-free inverses of linear operators (reference ``curvlinops/inverse.py:15-391``).

* ``CGInverseLinearOperator`` -- preconditioned conjugate gradients for symmetric positive
  definite operators.  The reference delegates to GPyTorch's ``linear_cg`` (a dependency that is
  not vendored); this is an own batched, device-resident implementation of the same method with the
  same keyword surface (``max_iter``, ``tolerance``, ``preconditioner``, ``initial_guess``, ``eps``;
  the tridiagonalisation options are accepted and ignored): all right-hand sides advance together,
  every iteration is one operator product plus a handful of fused vector updates on the device, and
  convergence is checked on the host only every few iterations.
* ``NeumannInverseLinearOperator`` -- truncated (optionally preconditioned) Neumann series.
* ``LSMRInverseLinearOperator`` -- SciPy's LSMR through ``.to_scipy()``, column by column.

The main use on this path: ``(G + delta I)^-1 v`` with the fast curvature matvec as ``A @`` and a
KFAC inverse as preconditioner.
"""

from __future__ import annotations

from collections.abc import Callable

import torch
from numpy import column_stack
from scipy.sparse.linalg import lsmr
from torch import Tensor, cat, isnan

from curvlinops_amd import _hip
from curvlinops_amd.linop import PyTorchLinearOperator
from curvlinops_amd.utils import is_native_tensor


class _InversePyTorchLinearOperator(PyTorchLinearOperator):
    """Base class: the inverse lives on the same (square) space as ``A``."""

    def __init__(self, A: PyTorchLinearOperator):
        if A._in_shape != A._out_shape:
            raise ValueError(
                f"Input linear operator must be square to form an inverse. Got {A._in_shape} != {A._out_shape}."
            )
        super().__init__(A._in_shape, A._out_shape)
        self._A = A

    @property
    def device(self) -> torch.device:
        return self._A.device

    @property
    def dtype(self) -> torch.dtype:
        return self._A.dtype

    def _flatten(self, X: list[Tensor]) -> Tensor:
        return cat([x.flatten(end_dim=-2) for x in X])

    def _unflatten(self, Y: Tensor) -> list[Tensor]:
        K = Y.shape[1]
        return [r.reshape(*s, K) for r, s in zip(Y.split(self._out_shape_flat), self._out_shape)]


def conjugate_gradients(
    matmul: Callable[[Tensor], Tensor],
    B: Tensor,
    max_iter: int | None = None,
    tolerance: float = 1e-6,
    preconditioner: Callable[[Tensor], Tensor] | None = None,
    initial_guess: Tensor | None = None,
    eps: float = 1e-30,
    check_every: int = 5,
) -> Tensor:
    """Solve ``A X = B`` column-wise for symmetric positive definite ``A`` given as ``matmul``.

    ``B`` is ``[D, K]``; a column is converged once ``||r|| <= tolerance * ||b||`` and is frozen
    from then on (its step sizes are zeroed), the loop stops when all columns are.  All reductions
    stay on the device; the host looks at the convergence flags every ``check_every`` iterations."""
    D, K = B.shape
    max_iter = min(D, 1000) if max_iter is None else max_iter
    if K == 1 and is_native_tensor(B) and (initial_guess is None or is_native_tensor(initial_guess)):
        x = _conjugate_gradients_native(matmul, B.reshape(-1), max_iter, tolerance, preconditioner,
                                        None if initial_guess is None else initial_guess.reshape(-1), check_every)
        return x.unsqueeze(1)
    if K == 1:  # single right-hand side: hand the operator a vector (its fastest format)
        mm, pc = matmul, preconditioner
        matmul = lambda P: mm(P.squeeze(1)).unsqueeze(1)  # noqa: E731
        if pc is not None:
            preconditioner = lambda R: pc(R.squeeze(1)).unsqueeze(1)  # noqa: E731
    X = torch.zeros_like(B) if initial_guess is None else initial_guess.clone()
    R = B - matmul(X) if initial_guess is not None else B.clone()
    Z = preconditioner(R) if preconditioner is not None else R
    P = Z.clone()
    rz = (R * Z).sum(dim=0)
    bnorm = B.norm(dim=0).clamp_min(eps)
    for it in range(max_iter):
        AP = matmul(P)
        pAp = (P * AP).sum(dim=0)
        active = (R.norm(dim=0) > tolerance * bnorm) & (pAp.abs() > eps)
        alpha = torch.where(active, rz / pAp.clamp_min(eps), torch.zeros_like(rz))
        X.addcmul_(P, alpha)
        R.addcmul_(AP, alpha, value=-1.0)
        Z = preconditioner(R) if preconditioner is not None else R
        rz_new = (R * Z).sum(dim=0)
        beta = torch.where(active, rz_new / rz.clamp_min(eps), torch.zeros_like(rz))
        P = Z + P * beta
        rz = rz_new
        if (it + 1) % check_every == 0 and not bool(((R.norm(dim=0) > tolerance * bnorm)).any()):
            break
    return X


def _conjugate_gradients_native(matmul, b: Tensor, max_iter: int, tolerance: float, preconditioner,
                                x0: Tensor | None, check_every: int) -> Tensor:
    """fp32 GPU, one right-hand side: per iteration ONE operator product, one dot and two fused update
    kernels (``clo_cg_update_f32``: x, r and <r, r> in one pass; ``clo_cg_direction_f32``); the step
    sizes are formed on the device, the host reads the residual norm every ``check_every`` iterations."""
    lib = _hip.load()
    dev, n = b.device, b.numel()
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = torch.empty(lib.clo_dot_ws_bytes(), device=dev, dtype=torch.uint8)
    scal = torch.zeros(4, device=dev, dtype=torch.float32)  # rz, pAp, rr / rz_new, spare
    rz, pap, rr, rz_new = (scal[i:i + 1] for i in range(4))
    ptr = lambda t: t.data_ptr()  # noqa: E731

    def dot(u: Tensor, v: Tensor, out: Tensor) -> None:
        rc = lib.clo_dot_f32(ptr(u), ptr(v), n, 1.0, ptr(out), ws.data_ptr(), stream)
        if rc:
            _hip._check(rc, "clo_dot_f32")

    x = torch.zeros_like(b) if x0 is None else x0.clone().contiguous()
    r = b.clone() if x0 is None else b - matmul(x)
    r = r.contiguous()
    z = preconditioner(r).contiguous() if preconditioner is not None else r
    p = z.clone()
    dot(r, z, rz)
    bnorm2 = float(torch.dot(b, b))
    thresh = (tolerance**2) * max(bnorm2, 1e-60)
    for it in range(max_iter):
        ap = matmul(p)
        ap = ap if ap.is_contiguous() else ap.contiguous()
        dot(p, ap, pap)
        rc = lib.clo_cg_update_f32(ptr(x), ptr(r), ptr(p), ptr(ap), n, ptr(rz), ptr(pap), ptr(rr), ws.data_ptr(),
                                   stream)
        if rc:
            _hip._check(rc, "clo_cg_update_f32")
        if preconditioner is not None:
            z = preconditioner(r).contiguous()
            dot(r, z, rz_new)
            num = rz_new
        else:
            z, num = r, rr
        rc = lib.clo_cg_direction_f32(ptr(p), ptr(z), n, ptr(num), ptr(rz), stream)
        if rc:
            _hip._check(rc, "clo_cg_direction_f32")
        rz.copy_(num)
        if (it + 1) % check_every == 0 and float(rr) <= thresh:
            break
    return x


class CGInverseLinearOperator(_InversePyTorchLinearOperator):
    """``A^-1`` of a symmetric positive definite operator by (preconditioned) conjugate gradients."""

    _IGNORED = ("n_tridiag", "max_tridiag_iter", "stop_updating_after")

    def __init__(self, A: PyTorchLinearOperator, **cg_hyperparameters):
        super().__init__(A)
        self.SELF_ADJOINT = A.SELF_ADJOINT
        unknown = set(cg_hyperparameters) - {"max_iter", "tolerance", "preconditioner", "initial_guess", "eps",
                                             *self._IGNORED}
        if unknown:
            raise TypeError(f"Unknown CG hyper-parameters: {sorted(unknown)}.")
        self._cg_hyperparameters = cg_hyperparameters

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        kw = {k: v for k, v in self._cg_hyperparameters.items() if k not in self._IGNORED}
        return self._unflatten(conjugate_gradients(self._A.__matmul__, self._flatten(X), **kw))

    def _adjoint(self) -> "CGInverseLinearOperator":
        return CGInverseLinearOperator(self._A.adjoint(), **self._cg_hyperparameters)


class LSMRInverseLinearOperator(_InversePyTorchLinearOperator):
    """``A^-1`` (least-squares sense) by SciPy's LSMR on the host, one column at a time."""

    def __init__(self, A: PyTorchLinearOperator, **lsmr_hyperparameters):
        super().__init__(A)
        self._A_scipy = A.to_scipy()
        self._lsmr_hyperparameters = lsmr_hyperparameters

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        X_np = self._flatten(X).cpu().numpy()
        cols = [lsmr(self._A_scipy, x, **self._lsmr_hyperparameters)[0] for x in X_np.T]
        Y = torch.as_tensor(column_stack(cols), dtype=self.dtype, device=self.device)
        return self._unflatten(Y)

    def _adjoint(self) -> "LSMRInverseLinearOperator":
        return LSMRInverseLinearOperator(self._A.adjoint(), **self._lsmr_hyperparameters)


class NeumannInverseLinearOperator(_InversePyTorchLinearOperator):
    r"""Truncated Neumann series :math:`A^{-1} \approx \alpha \sum_{k=0}^{K} (I - \alpha P A)^k P`
    (``P`` = optional left preconditioner, :math:`\alpha` = ``scale``)."""

    def __init__(self, A: PyTorchLinearOperator, num_terms: int = 100, scale: float = 1.0,
                 check_nan: bool = True, preconditioner: Callable[[Tensor], Tensor] | None = None):
        super().__init__(A)
        self._num_terms, self._scale, self._check_nan = num_terms, scale, check_nan
        self._preconditioner = preconditioner

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        if self._preconditioner is None:
            rhs, step = X, self._A._matmat
        else:
            def P(Xl: list[Tensor]) -> list[Tensor]:
                return self._unflatten(self._preconditioner(self._flatten(Xl)))

            rhs = P(X)

            def step(v: list[Tensor]) -> list[Tensor]:
                return P(self._A._matmat(v))

        result = [x.clone() for x in rhs]
        v = [x.clone() for x in rhs]
        for idx in range(self._num_terms):
            Av = step(v)
            v = [vi.sub_(Avi, alpha=self._scale) for vi, Avi in zip(v, Av)]
            result = [r.add_(vi) for r, vi in zip(result, v)]
            if self._check_nan and any(bool(isnan(r).any()) for r in result):
                raise ValueError(
                    f"Detected NaNs after application of {idx}-th term. This is probably because the "
                    "Neumann series is non-convergent. Try decreasing `scale`."
                )
        return [r.mul_(self._scale) for r in result]

    def _adjoint(self) -> "NeumannInverseLinearOperator":
        preconditioner = None
        if self._preconditioner is not None:
            owner = getattr(self._preconditioner, "__self__", None)
            if not isinstance(owner, PyTorchLinearOperator):
                raise NotImplementedError(
                    "Adjoint with a preconditioner is only supported when the preconditioner is a bound "
                    "PyTorchLinearOperator.__matmul__ method."
                )
            preconditioner = owner.adjoint().__matmul__
        return NeumannInverseLinearOperator(self._A.adjoint(), num_terms=self._num_terms, scale=self._scale,
                                            check_nan=self._check_nan, preconditioner=preconditioner)


__all__ = ["CGInverseLinearOperator", "LSMRInverseLinearOperator", "NeumannInverseLinearOperator",
           "conjugate_gradients"]
