"""Backpropagated vectors that realise the (approximate) Fisher of a loss for ONE datum:
columns of a loss-Hessian square root (type-2), sampled would-be gradients (MC), the actual
gradient (empirical) or nothing (forward-only).

Math restated from the reference (``curvlinops/ggn_utils.py:29-376``):
  MSE   S = sqrt(2c) I                               sample  N(0, 2c)
  CE    S = sqrt(c) (diag(sqrt p) - p sqrt(p)^T)     sample  sqrt(c) (p - onehot(y~p))
  BCE   S = sqrt(c) diag(sqrt(s (1-s)))              sample  sqrt(c) (s - y~Bernoulli(s))
with c = 1 ('sum') or 1 / #loss-terms-of-the-datum ('mean').  All functions act on a single
datum (no batch axis) and are vmapped by the callers.
"""

from __future__ import annotations

from collections.abc import Callable
from math import sqrt

import torch
from torch import Generator, Tensor
from torch.func import grad
from torch.nn import BCEWithLogitsLoss, CrossEntropyLoss, MSELoss
from torch.nn.functional import one_hot

from curvlinops_amd.enums import FisherType
from curvlinops_amd.utils import make_functional_loss

_LOSSES = (MSELoss, CrossEntropyLoss, BCEWithLogitsLoss)


def _datum_reduction(out: Tensor, loss_func) -> float:
    terms = out.numel() / out.shape[0] if isinstance(loss_func, CrossEntropyLoss) else out.numel()
    return {"sum": 1.0, "mean": 1.0 / terms}[loss_func.reduction]


def loss_hessian_matrix_sqrt(out: Tensor, target: Tensor, loss_func) -> Tensor:
    """Square root ``S`` with ``S S^T = nabla^2_out l``; shape ``[*out.shape, *out.shape]``."""
    c = _datum_reduction(out, loss_func)
    if isinstance(loss_func, MSELoss):
        S = torch.full_like(out, sqrt(2 * c)).flatten().diag()
    elif isinstance(loss_func, CrossEntropyLoss):
        flat = out.unsqueeze(-1).flatten(start_dim=1)  # [C, D]
        C, D = flat.shape
        p = flat.softmax(dim=0)
        ps = sqrt(c) * p.sqrt()
        # per position d: diag(ps_d) - p_d ps_d^T ; assembled in the (c, d) basis
        blocks = torch.diag_embed(ps.T) - p.T.unsqueeze(2) * ps.T.unsqueeze(1)  # [D, C, C]
        eye = torch.eye(D, dtype=out.dtype, device=out.device)
        S = torch.einsum("dab,de->adbe", blocks, eye).reshape(C * D, C * D)
    elif isinstance(loss_func, BCEWithLogitsLoss):
        s = out.flatten().sigmoid()
        S = (sqrt(c) * (s * (1 - s)).sqrt()).diag()
    else:
        raise NotImplementedError(f"Loss function {loss_func} not supported.")
    return S.reshape(*out.shape, *out.shape)


def _sample(out: Tensor, num: int, loss_func, generator: Generator | None) -> Tensor:
    """``num`` would-be gradients of one datum, shape ``[num, *out.shape]``."""
    c = _datum_reduction(out, loss_func)
    if isinstance(loss_func, MSELoss):
        mean = torch.zeros(num, *out.shape, device=out.device, dtype=out.dtype)
        std = torch.as_tensor(sqrt(2 * c), device=out.device, dtype=out.dtype)
        return torch.normal(mean, std, generator=generator)
    if isinstance(loss_func, CrossEntropyLoss):
        C = out.shape[0]
        p = out.unsqueeze(-1).flatten(start_dim=1).softmax(dim=0)  # [C, S]
        draws = p.T.multinomial(num_samples=num, replacement=True, generator=generator)  # [S, num]
        hot = one_hot(draws.T, num_classes=C).movedim(-1, 1)  # [num, C, S]
        return (sqrt(c) * (p.unsqueeze(0) - hot)).reshape(num, *out.shape)
    if isinstance(loss_func, BCEWithLogitsLoss):
        s = out.sigmoid().unsqueeze(0).expand(num, *out.shape)
        return sqrt(c) * (s - s.bernoulli(generator=generator))
    raise NotImplementedError(f"Supported losses: {_LOSSES}")


def make_grad_output_fn(loss_func, fisher_type: FisherType, mc_samples: int = 1) -> Callable[
    [Tensor, Tensor, Generator | None], Tensor
]:
    """``(output, target, generator) -> [V, *output.shape]`` for one datum; V = C (type-2),
    ``mc_samples`` (scaled by 1/sqrt(M)), 1 (empirical) or 0 (forward-only)."""
    if fisher_type not in FisherType:
        raise ValueError(f"Invalid fisher_type {fisher_type!r}. Must be one of {list(FisherType)}.")
    if fisher_type == FisherType.EMPIRICAL:
        c = make_functional_loss(loss_func)

        def datum_loss(pred: Tensor, target: Tensor) -> Tensor:
            (C,) = pred.shape
            mean_over_features = isinstance(loss_func, (BCEWithLogitsLoss, MSELoss)) and loss_func.reduction == "mean"
            return (sqrt(C) if mean_over_features else 1.0) * c(pred.unsqueeze(0), (target.unsqueeze(0),))

        datum_grad = grad(datum_loss, argnums=0)

    def fn(output: Tensor, target: Tensor, generator: Generator | None = None) -> Tensor:
        if fisher_type == FisherType.FORWARD_ONLY:
            return output.new_empty(0, *output.shape)
        if fisher_type == FisherType.TYPE2:
            S = loss_hessian_matrix_sqrt(output, target, loss_func)
            return S.reshape(*output.shape, output.numel()).movedim(-1, 0)
        if fisher_type == FisherType.MC:
            return _sample(output, mc_samples, loss_func, generator).div_(sqrt(mc_samples))
        return datum_grad(output, target).unsqueeze(0)

    return fn
