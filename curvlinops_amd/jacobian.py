"""Jacobian and transposed-Jacobian linear operators (reference ``curvlinops/jacobian.py:14-358``).

``JacobianLinearOperator @ v`` is the forward + JVP half of the GGN product, its adjoint the VJP
half; both are exposed with the reference's constructor signature.  The output space is the
CONCATENATION over mini-batches (``[(N_data, *out_shape)]``), so the data order must be fixed.

Execution paths, as for the curvature operators:

* **native** -- fully-connected nets in fp32 on the GPU: ``clo_mlp_jvp`` (one fused tangent-forward
  launch per layer) and ``clo_mlp_vjp`` (forward + backward chain on the GEMM engine);
* **autograd** -- everything else: ``torch.func.jvp`` / ``vjp`` vmapped over the trailing column axis.
"""

from __future__ import annotations

from collections.abc import Callable, Iterable, MutableMapping

import torch
from torch import Tensor
from torch.func import jvp, vjp, vmap
from torch.nn import Module

from curvlinops_amd import _hip
from curvlinops_amd.curvature import CurvatureLinearOperator
from curvlinops_amd.mlp_native import NativeMLP, detect_mlp


def make_batch_jacobian_matrix_product(f: Callable) -> Callable:
    """``(params, X, M_dict) -> J M`` with the columns of ``M`` on the trailing axis."""

    @torch.no_grad()
    def jacobian_vector_product(params, X, v):
        return jvp(lambda p: f(p, X), (params,), (v,))[1]

    return vmap(jacobian_vector_product, in_dims=(None, None, -1), out_dims=-1, randomness="same")


def make_batch_transposed_jacobian_matrix_product(f: Callable) -> Callable:
    """``(params, X, U) -> J^T U`` as a dict, columns of ``U`` on the trailing axis."""

    @torch.no_grad()
    def transposed_jacobian_vector_product(params, X, u):
        (result,) = vjp(lambda p: f(p, X), params)[1](u)
        return result

    return vmap(transposed_jacobian_vector_product, in_dims=(None, None, -1), out_dims=-1, randomness="same")


class _JacobianBase(CurvatureLinearOperator):
    FIXED_DATA_ORDER: bool = True

    def __init__(
        self,
        model_func: Module | Callable[[dict[str, Tensor], Tensor | MutableMapping], Tensor],
        params: dict[str, Tensor],
        data: Iterable[tuple[Tensor | MutableMapping, Tensor]],
        progressbar: bool = False,
        check_deterministic: bool = True,
        num_data: int | None = None,
        batch_size_fn: Callable[[Tensor | MutableMapping], int] | None = None,
    ):
        super().__init__(model_func, None, params, data, progressbar=progressbar,
                         check_deterministic=check_deterministic, num_data=num_data,
                         batch_size_fn=batch_size_fn)

    def _output_space(self) -> list[tuple[int, ...]]:
        x = next(iter(self._data))[0]
        if isinstance(x, Tensor):
            x = x.to(self.device)
        with torch.no_grad():
            out = self._model_func(self._params, x)
        return [(self._N_data, *out.shape[1:])]

    def _init_native(self) -> None:
        structure = detect_mlp(self._model_module, self._params)
        if structure is None:
            return
        _hip.load()  # a GPU fp32 MLP must run natively: fail loudly if the library is absent
        self._native = NativeMLP(structure, self._params)

    def _native_batches(self) -> list[Tensor] | None:
        """The prepared inputs of all mini-batches, or None if one does not qualify."""
        out = []
        for X, _ in self._loop_over_data(desc="_matmat"):
            Xn = self._native.prepare_input(X)
            if Xn is None or Xn.shape[0] == 0:
                return None
            out.append(Xn)
        return out


class JacobianLinearOperator(_JacobianBase):
    r"""The model's Jacobian :math:`\mathbf{J}_\theta \mathbf{f}`, an :math:`NC \times D` matrix with
    rows ordered by datum, then output entry."""

    def _init_mp(self) -> None:
        self._mp = make_batch_jacobian_matrix_product(self._model_func)

    def _get_out_shape(self) -> list[tuple[int, ...]]:
        return self._output_space()

    def _matmat(self, M: list[Tensor]) -> list[Tensor]:
        self._native_rebind_if_replaced()   # params are held by reference: follow swapped storage
        if self._native is not None and all(m.is_cuda and m.dtype == torch.float32 for m in M):
            out = self._matmat_native(M)
            if out is not None:
                return out
        M_dict = dict(zip(self._params.keys(), M))
        return [torch.cat([self._mp(self._params, X, M_dict) for X, _ in self._loop_over_data(desc="_matmat")])]

    def _matmat_native(self, M: list[Tensor]) -> list[Tensor] | None:
        nat = self._native
        batches = self._native_batches()
        if batches is None:
            return None
        K = M[0].shape[-1]
        C = nat.s.dims[-1]
        Vk = [m.movedim(-1, 0).contiguous() for m in M]  # K-major: every column parameter-shaped
        out = torch.empty(K, self._N_data, C, device=self.device, dtype=torch.float32)
        for k in range(K):
            VW = [Vk[i][k] for i in nat.w_idx]
            Vb = [None if i is None else Vk[i][k] for i in nat.b_idx]
            row = 0
            for Xn in batches:
                n = Xn.shape[0]
                nat.plan.jvp(nat.W, nat.b, VW, Vb, Xn, out[k, row:row + n])
                row += n
        return [out.movedim(0, -1).reshape(*self._out_shape[0], K)]

    def _adjoint(self) -> "TransposedJacobianLinearOperator":
        return TransposedJacobianLinearOperator(
            self._model_func, self._params, self._data, progressbar=self._progressbar,
            check_deterministic=False, batch_size_fn=self._batch_size_fn, num_data=self._N_data)


class TransposedJacobianLinearOperator(_JacobianBase):
    r"""The transposed Jacobian :math:`(\mathbf{J}_\theta \mathbf{f})^\top`, :math:`D \times NC`."""

    def _init_mp(self) -> None:
        self._mp = make_batch_transposed_jacobian_matrix_product(self._model_func)

    def _get_in_shape(self) -> list[tuple[int, ...]]:
        return self._output_space()

    def _matmat(self, M: list[Tensor]) -> list[Tensor]:
        self._native_rebind_if_replaced()   # params are held by reference: follow swapped storage
        if self._native is not None and all(m.is_cuda and m.dtype == torch.float32 for m in M):
            out = self._matmat_native(M)
            if out is not None:
                return out
        (num_vectors,) = {m.shape[-1] for m in M}
        JTM = {name: p.new_zeros(*p.shape, num_vectors) for name, p in self._params.items()}
        processed = 0
        for X, _ in self._loop_over_data(desc="_matmat"):
            n = self._batch_size_fn(X)
            for name, val in self._mp(self._params, X, M[0][processed:processed + n]).items():
                JTM[name].add_(val)
            processed += n
        return list(JTM.values())

    def _matmat_native(self, M: list[Tensor]) -> list[Tensor] | None:
        nat = self._native
        batches = self._native_batches()
        if batches is None:
            return None
        K = M[0].shape[-1]
        C = nat.s.dims[-1]
        Uk = M[0].reshape(self._N_data, C, K).movedim(-1, 0).contiguous()  # [K, N_data, C]
        Ok = [torch.empty(K, *p.shape, device=self.device, dtype=torch.float32) for p in self._params.values()]
        for k in range(K):
            OW = [Ok[i][k] for i in nat.w_idx]
            Ob = [None if i is None else Ok[i][k] for i in nat.b_idx]
            row = 0
            for bi, Xn in enumerate(batches):
                n = Xn.shape[0]
                nat.plan.vjp(nat.W, nat.b, OW, Ob, Xn, Uk[k, row:row + n], 1.0, 0.0 if bi == 0 else 1.0)
                row += n
        return [o.movedim(0, -1) for o in Ok]

    def _adjoint(self) -> JacobianLinearOperator:
        return JacobianLinearOperator(
            self._model_func, self._params, self._data, progressbar=self._progressbar,
            check_deterministic=False, batch_size_fn=self._batch_size_fn, num_data=self._N_data)


__all__ = ["JacobianLinearOperator", "TransposedJacobianLinearOperator"]
