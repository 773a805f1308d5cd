"""Empirical-risk plumbing shared by curvature operators and KFAC computers: the data loop,
data-set statistics, normalisation factors and the determinism guard.

Semantics follow the reference's ``curvlinops/_empirical_risk.py:20-439`` (constructor
contract, ``TypeError`` for non-dict params, ``B_b / N_data`` normalisation for
``reduction='mean'``, two-pass determinism check with ``rtol=5e-5, atol=1e-6``).
"""

from __future__ import annotations

from collections.abc import Callable, Iterable, Iterator, MutableMapping

import torch
from torch import Tensor
from torch.nn import CrossEntropyLoss, Module

from curvlinops_amd.utils import (
    allclose_report,
    enable_requires_grad,
    infer_device,
    infer_dtype,
    make_functional_call,
)


class EmpiricalRiskMixin:
    """State and helpers of an object that sweeps a data set of mini-batches ``(X, y)``.

    Attributes:
        FIXED_DATA_ORDER: the determinism guard additionally compares batch by batch.
        NEEDS_NUM_PER_EXAMPLE_LOSS_TERMS: infer the number of loss terms per datum.
    """

    FIXED_DATA_ORDER: bool = False
    NEEDS_NUM_PER_EXAMPLE_LOSS_TERMS: bool = False

    def __init__(
        self,
        model_func: Module | Callable[[dict[str, Tensor], Tensor | MutableMapping], Tensor],
        loss_func: Callable[[Tensor, Tensor], Tensor] | None,
        params: dict[str, Tensor],
        data: Iterable[tuple[Tensor | MutableMapping, Tensor]],
        progressbar: bool = False,
        batch_size_fn: Callable[[MutableMapping | Tensor], int] | None = None,
        num_data: int | None = None,
        num_per_example_loss_terms: int | None = None,
        check_deterministic: bool = True,
    ):
        first = next(iter(data), None)  # None: an empty shard of a data-parallel run (needs num_data and
        #                                   num_per_example_loss_terms from the caller)
        if first is not None and isinstance(first[0], MutableMapping) and batch_size_fn is None:
            raise ValueError("When using dict-like custom data, `batch_size_fn` is required.")
        if not isinstance(params, dict):
            raise TypeError(
                f"params must be a dict[str, Tensor], got {type(params).__name__}. "
                "Use dict(model.named_parameters()) instead of list(model.parameters())."
            )
        if isinstance(model_func, Module):
            self._model_module: Module | None = model_func
            self._model_func = make_functional_call(model_func)
        elif callable(model_func):
            self._model_module = None
            self._model_func = model_func
        else:
            raise ValueError(
                f"model_func must be an nn.Module or a callable, got {type(model_func).__name__}."
            )
        self._params = params
        self._loss_func = loss_func
        self._data = data
        self._progressbar = progressbar
        self._batch_size_fn = (lambda X: X.shape[0]) if batch_size_fn is None else batch_size_fn
        self._N_data, self._num_per_example_loss_terms = self._data_statistics(
            num_data, num_per_example_loss_terms
        )
        if check_deterministic:
            self._check_deterministic()

    # ------------------------------------------------------------------ properties
    @property
    def device(self) -> torch.device:
        return infer_device(self._params.values())

    @property
    def dtype(self) -> torch.dtype:
        return infer_dtype(self._params.values())

    # ------------------------------------------------------------------ data loop
    def _loop_over_data(self, desc: str | None = None) -> Iterator[tuple[Tensor | MutableMapping, Tensor]]:
        """Yield mini-batches moved to the operator's device (host->device boundary)."""
        it = self._data
        dev = self.device
        if self._progressbar:
            from tqdm import tqdm

            label = f"{self.__class__.__name__}{'' if desc is None else f'.{desc}'} (on {dev})"
            it = tqdm(it, desc=label)
        for X, y in it:
            if isinstance(X, Tensor):
                X = X.to(dev)
            yield X, y.to(dev)

    def _get_normalization_factor(self, X, y: Tensor) -> float:
        """1 for ``reduction='sum'``, ``B_b / N_data`` for ``'mean'``."""
        return {"sum": 1.0, "mean": self._batch_size_fn(X) / self._N_data}[self._loss_func.reduction]

    def _data_statistics(self, num_data, num_per_example_loss_terms):
        need_n = num_data is None
        need_terms = (
            self.NEEDS_NUM_PER_EXAMPLE_LOSS_TERMS
            and self._loss_func is not None
            and num_per_example_loss_terms is None
        )
        if not need_n and not need_terms:
            return num_data, num_per_example_loss_terms
        n_acc, t_acc = 0, 0
        for X, y in self._loop_over_data(desc="data_statistics"):
            n_acc += self._batch_size_fn(X)
            if need_terms:
                t_acc += y.numel() if isinstance(self._loss_func, CrossEntropyLoss) else y.shape[:-1].numel()
        N = n_acc if need_n else num_data
        if need_terms:
            # terms per datum from the data actually seen (a rank's shard when num_data is the
            # global count of a sharded data set)
            if n_acc == 0 or t_acc % n_acc != 0:
                raise ValueError(
                    "The number of loss terms must be divisible by the number of data points; "
                    f"num_loss_terms={t_acc}, N_data={n_acc}."
                )
            num_per_example_loss_terms = t_acc // n_acc
        return N, num_per_example_loss_terms

    # ------------------------------------------------------------------ loss / gradient sweeps
    def _batch_prediction_loss_gradient(self):
        """Yield ``((X, y), prediction, loss, [grads])`` per batch (normalised loss)."""
        plist = list(self._params.values())
        for X, y in self._loop_over_data(desc="batch_prediction_loss_gradient"):
            if self._loss_func is None:
                with torch.no_grad():  # (never yield inside: the grad mode would leak to the caller)
                    pred = self._model_func(self._params, X).detach()
                yield (X, y), pred, None, None
                continue
            with enable_requires_grad(plist):
                pred = self._model_func(self._params, X)
                loss = self._loss_func(pred, y) * self._get_normalization_factor(X, y)
                grads = torch.autograd.grad(loss, plist)
            yield (X, y), pred.detach(), loss.detach(), [g.detach() for g in grads]

    def _gradient_and_loss(self) -> tuple[list[Tensor], Tensor]:
        if self._loss_func is None:
            raise ValueError("No loss function specified.")
        total_loss = torch.zeros((), device=self.device, dtype=self.dtype)
        total_grad = [torch.zeros_like(p) for p in self._params.values()]
        for _, _, loss, grads in self._batch_prediction_loss_gradient():
            total_loss.add_(loss)
            for t, g in zip(total_grad, grads):
                t.add_(g)
        return total_grad, total_loss

    def _check_deterministic(self, rtol: float = 5e-5, atol: float = 1e-6) -> None:
        """Two sweeps over the data must agree in total loss and gradient (and batch by batch
        when ``FIXED_DATA_ORDER``); ``RuntimeError`` otherwise."""
        has_loss = self._loss_func is not None
        if has_loss:
            g1 = [torch.zeros_like(p) for p in self._params.values()]
            g2 = [torch.zeros_like(p) for p in self._params.values()]
            l1 = torch.zeros((), device=self.device, dtype=self.dtype)
            l2 = torch.zeros((), device=self.device, dtype=self.dtype)
        for first, second in zip(self._batch_prediction_loss_gradient(), self._batch_prediction_loss_gradient()):
            (X1, y1), p1, b1, gr1 = first
            (X2, y2), p2, b2, gr2 = second
            if self.FIXED_DATA_ORDER:
                self._compare_batches((X1, X2), (y1, y2), (p1, p2), (b1, b2), (gr1, gr2), has_loss, rtol, atol)
            if has_loss:
                l1.add_(b1)
                l2.add_(b2)
                for t, g in zip(g1, gr1):
                    t.add_(g)
                for t, g in zip(g2, gr2):
                    t.add_(g)
        if has_loss:
            if not allclose_report(l1, l2, rtol=rtol, atol=atol):
                raise RuntimeError("Check for deterministic total loss failed.")
            if any(not allclose_report(a, b, rtol=rtol, atol=atol) for a, b in zip(g1, g2)):
                raise RuntimeError("Check for deterministic total gradient failed.")

    @staticmethod
    def _compare_batches(Xs, ys, preds, losses, grads, has_loss, rtol, atol) -> None:
        X1, X2 = Xs
        if isinstance(X1, MutableMapping) and isinstance(X2, MutableMapping):
            for k in X1:
                if isinstance(X1[k], Tensor) and not allclose_report(X1[k], X2[k], rtol=rtol, atol=atol):
                    raise RuntimeError("Check for deterministic X failed.")
        elif not allclose_report(X1, X2, rtol=rtol, atol=atol):
            raise RuntimeError("Check for deterministic X failed.")
        if not allclose_report(ys[0], ys[1], rtol=rtol, atol=atol):
            raise RuntimeError("Check for deterministic y failed.")
        if not allclose_report(preds[0], preds[1], rtol=rtol, atol=atol):
            raise RuntimeError("Check for deterministic batch prediction failed.")
        if has_loss:
            if not allclose_report(losses[0], losses[1], rtol=rtol, atol=atol):
                raise RuntimeError("Check for deterministic batch loss failed.")
            if any(not allclose_report(a, b, rtol=rtol, atol=atol) for a, b in zip(grads[0], grads[1])):
                raise RuntimeError("Check for deterministic batch gradient failed.")
