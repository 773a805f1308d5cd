"""Diagonal of the generalized Gauss-Newton matrix as a linear operator (reference
``curvlinops/ggn_diagonal.py:8-91`` and ``computers/ggn_diagonal.py:21-232``).

Per mini-batch the diagonal is ``scale * sum_n sum_v (grad_theta <g_nv, f_n>)^2`` with ``g_nv`` the
columns of the loss Hessian's square root (exact, ``mc_samples = 0``) or would-be gradients sampled
from the model's likelihood.  Two execution paths:

* nets made of ``Linear`` / ``Conv2d`` layers: hooks + the fused squared-per-example-gradient kernel
  (:class:`curvlinops_amd.computers.HipGGNDiagonalComputer`);
* everything else: ``torch.func`` (``vmap`` over data of a ``vjp`` per backpropagated vector).
"""

from __future__ import annotations

from collections.abc import Callable, Iterable, MutableMapping

import torch
from torch import Tensor
from torch.func import vjp, vmap
from torch.nn import Module

from curvlinops_amd.computers import HipGGNDiagonalComputer
from curvlinops_amd.diag import DiagonalLinearOperator
from curvlinops_amd.enums import FisherType
from curvlinops_amd.loss_sampling import make_grad_output_fn
from curvlinops_amd.risk import EmpiricalRiskMixin
from curvlinops_amd.utils import seed_generator


class _FuncGGNDiagonalComputer(EmpiricalRiskMixin):
    """General nets: per-datum ``vjp`` of every backpropagated vector, squared and summed."""

    def __init__(self, model_func, loss_func, params, data, progressbar=False, check_deterministic=True,
                 num_data=None, batch_size_fn=None, mc_samples: int = 0, seed: int = 2_147_483_647):
        self._mc_samples, self._seed = mc_samples, seed
        if mc_samples > 0:
            self.FIXED_DATA_ORDER = True
        super().__init__(model_func, loss_func, params, data, progressbar=progressbar,
                         batch_size_fn=batch_size_fn, num_data=num_data, check_deterministic=check_deterministic)

    def compute(self) -> dict[str, Tensor]:
        fisher = FisherType.TYPE2 if self._mc_samples == 0 else FisherType.MC
        grad_output_fn = make_grad_output_fn(self._loss_func, fisher, max(self._mc_samples, 1))
        f, params = self._model_func, self._params

        def datum(x, y, generator):
            f_x, f_vjp = vjp(lambda p: f(p, x.unsqueeze(0)).squeeze(0), params)
            (g,) = vmap(f_vjp)(grad_output_fn(f_x.detach(), y, generator))
            return {k: (g[k] ** 2).sum(0) for k in params}

        batched = vmap(datum, in_dims=(0, 0, None), randomness="different" if self._mc_samples else "same")
        generator = None if self._mc_samples == 0 else seed_generator(None, self.device, self._seed)
        result = {k: torch.zeros_like(p) for k, p in params.items()}
        with torch.no_grad():
            for X, y in self._loop_over_data(desc="GGN diagonal"):
                scale = {"sum": 1.0, "mean": 1.0 / self._batch_size_fn(X)}[self._loss_func.reduction]
                for k, v in batched(X, y, generator).items():
                    result[k].add_(v.sum(0), alpha=scale * self._get_normalization_factor(X, y))
        return result


class GGNDiagonalLinearOperator(DiagonalLinearOperator):
    """``diag(G)`` of the GGN (exact) or of its Monte-Carlo approximation (``mc_samples > 0``)."""

    def __init__(
        self,
        model_func: Module | Callable[[dict[str, Tensor], Tensor | MutableMapping], Tensor],
        loss_func: Callable[[Tensor, Tensor], Tensor],
        params: dict[str, Tensor],
        data: Iterable[tuple[Tensor | MutableMapping, Tensor]],
        progressbar: bool = False,
        check_deterministic: bool = True,
        num_data: int | None = None,
        batch_size_fn: Callable[[MutableMapping | Tensor], int] | None = None,
        mc_samples: int = 0,
        seed: int = 2_147_483_647,
    ):
        diagonal = None
        if isinstance(model_func, Module):
            try:
                comp = HipGGNDiagonalComputer(
                    model_func, loss_func, params, data, progressbar=progressbar,
                    check_deterministic=check_deterministic, seed=seed,
                    fisher_type=FisherType.TYPE2 if mc_samples == 0 else FisherType.MC,
                    mc_samples=max(mc_samples, 1), separate_weight_and_bias=False, num_data=num_data,
                    batch_size_fn=batch_size_fn)
                diagonal = comp.compute()
            except (NotImplementedError, ValueError):
                diagonal = None  # parameters / loss outside the hooks path: general route below
        if diagonal is None:
            diagonal = _FuncGGNDiagonalComputer(
                model_func, loss_func, params, data, progressbar=progressbar,
                check_deterministic=check_deterministic, num_data=num_data, batch_size_fn=batch_size_fn,
                mc_samples=mc_samples, seed=seed).compute()
        super().__init__([diagonal[k] for k in params])


__all__ = ["GGNDiagonalLinearOperator"]
