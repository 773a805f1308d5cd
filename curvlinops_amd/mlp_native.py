"""Recognise fully-connected nets and drive the native (HIP) GGN/EF matvec for them.

A model qualifies if it is an ``nn.Sequential`` of ``Linear`` layers, each optionally followed by
one of ReLU / Tanh / Sigmoid / Identity (leading ``Flatten`` allowed), all trainable parameters are
in ``params``, everything is fp32 on one GPU, the loss is MSE / CrossEntropy / BCEWithLogits and
the inputs are 2-D.  The matvec then never touches autograd: per mini-batch one call into
``clo_mlp_ggn_matvec`` (see ``csrc/mlp.hip``), which replaces the reference's
``vmap(jvp -> jvp(jacrev(c)) -> vjp)`` (``curvlinops/ggn.py:41-72``).
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch
from torch import Tensor, nn

from curvlinops_amd import _hip

_ACT_CODES = {nn.ReLU: _hip.ACT_RELU, nn.Tanh: _hip.ACT_TANH, nn.Sigmoid: _hip.ACT_SIGMOID,
              nn.Identity: _hip.ACT_IDENTITY}


@dataclass
class MLPStructure:
    dims: list[int]
    acts: list[int]
    weight_names: list[str]
    bias_names: list[str | None]
    leading_flatten: bool = False
    extra: dict = field(default_factory=dict)


def detect_mlp(model: nn.Module | None, params: dict[str, Tensor]) -> MLPStructure | None:
    """Return the layer table if ``model`` is a supported fully-connected stack whose Linear
    parameters are exactly ``params``; else None."""
    if not isinstance(model, nn.Sequential) or len(model) == 0:
        return None
    # every POSITION of the container (``named_children`` de-duplicates repeated instances, which would
    # silently drop a shared activation / layer); a module that appears twice is not supported natively
    mods = list(model._modules.items())
    if any(m is None for _, m in mods) or len({id(m) for _, m in mods}) != len(mods):
        return None
    leading_flatten = False
    if isinstance(mods[0][1], nn.Flatten):
        if mods[0][1].start_dim != 1 or mods[0][1].end_dim != -1:
            return None
        leading_flatten = True
        mods = mods[1:]
    dims: list[int] = []
    acts: list[int] = []
    wn: list[str] = []
    bn: list[str | None] = []
    i = 0
    while i < len(mods):
        name, m = mods[i]
        if not isinstance(m, nn.Linear):
            return None
        if dims and dims[-1] != m.in_features:
            return None
        if not dims:
            dims.append(m.in_features)
        dims.append(m.out_features)
        wn.append(f"{name}.weight")
        bn.append(f"{name}.bias" if m.bias is not None else None)
        act = _hip.ACT_IDENTITY
        if i + 1 < len(mods) and type(mods[i + 1][1]) in _ACT_CODES:
            act = _ACT_CODES[type(mods[i + 1][1])]
            i += 1
        acts.append(act)
        i += 1
    if not wn:
        return None
    expected = set(wn) | {b for b in bn if b is not None}
    if set(params.keys()) != expected:
        return None
    for n in expected:
        p = params[n]
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
            return None
    return MLPStructure(dims, acts, wn, bn, leading_flatten)


def loss_kind_and_scale(loss_func: nn.Module, N: int, C: int) -> tuple[int, float] | None:
    """(CLO_LOSS_* code, Hessian scale) of the exact loss Hessian for a batch of N rows with C
    outputs: MSE ``2c I``, CE ``c (diag p - p p^T)``, BCE ``c diag(s(1-s))`` with the
    reduction factor ``c`` (reference ``ggn_utils.py:44-79``)."""
    red = getattr(loss_func, "reduction", None)
    if red not in ("mean", "sum"):
        return None
    if isinstance(loss_func, nn.MSELoss):
        return _hip.LOSS_MSE, 2.0 * (1.0 / (N * C) if red == "mean" else 1.0)
    if isinstance(loss_func, nn.CrossEntropyLoss):
        if loss_func.weight is not None or loss_func.label_smoothing != 0.0 or loss_func.ignore_index >= 0:
            return None
        return _hip.LOSS_CE, (1.0 / N if red == "mean" else 1.0)
    if isinstance(loss_func, nn.BCEWithLogitsLoss):
        if loss_func.weight is not None or loss_func.pos_weight is not None:
            return None
        return _hip.LOSS_BCE, (1.0 / (N * C) if red == "mean" else 1.0)
    return None


class NativeMLP:
    """Argument marshalling for ``clo_mlp_ggn_matvec`` bound to one (model, params) pair."""

    def __init__(self, structure: MLPStructure, params: dict[str, Tensor]):
        self.s = structure
        self.plan = _hip.MLPPlan(structure.dims, structure.acts)
        self.names = list(params.keys())
        self.index = {n: i for i, n in enumerate(self.names)}
        self.W = [params[n] for n in structure.weight_names]
        self.b = [None if n is None else params[n] for n in structure.bias_names]
        self.w_idx = [self.index[n] for n in structure.weight_names]
        self.b_idx = [None if n is None else self.index[n] for n in structure.bias_names]
        # byte offsets of every parameter inside a flat [D] vector (params order)
        offs, pos = {}, 0
        for n in self.names:
            offs[n] = 4 * pos
            pos += params[n].numel()
        self.D = pos
        self.w_off = [offs[n] for n in structure.weight_names]
        self.b_off = [None if n is None else offs[n] for n in structure.bias_names]
        self.bound = [params[n] for n in self.names]  # the tensor objects the plan's pointers refer to
        self.bound_ptrs = [p.data_ptr() for p in self.bound]
        self.bound_shapes = [p.shape for p in self.bound]
        self.plan.bind_params(self.W, self.b)

    def is_current(self, params: dict[str, Tensor]) -> bool:
        """True if the plan's pointer tables still describe ``params``: same tensor objects AND same storage
        addresses, dtype and shape.  ``p.data = t``, ``vector_to_parameters``, ``module.to(...)`` keep the
        ``Parameter`` object and swap its storage, so object identity alone says nothing (the reference reads
        ``params`` afresh on every product, ``_torch_base.py:923-944``)."""
        if len(params) != len(self.bound):
            return False
        for p, q, ptr, shape in zip(params.values(), self.bound, self.bound_ptrs, self.bound_shapes):
            if p is not q or p.data_ptr() != ptr or p.dtype is not torch.float32 or p.shape != shape:
                return False
        return True

    def prepare_input(self, X: Tensor) -> Tensor | None:
        if not isinstance(X, Tensor) or not X.is_cuda or X.dtype != torch.float32:
            return None
        if self.s.leading_flatten:
            X = X.flatten(1)
        if X.dim() != 2 or X.shape[1] != self.s.dims[0]:
            return None
        return X.contiguous()

    def matvec(self, V: list[Tensor], out: list[Tensor], X: Tensor, loss_kind: int, loss_scale: float,
               alpha: float, beta: float, aux: Tensor | None = None) -> None:
        """``out = beta*out + alpha * (J^T H J) V`` for one mini-batch; ``V``/``out`` are
        parameter-shaped contiguous fp32 tensors in ``params`` order."""
        VW = [V[i] for i in self.w_idx]
        Vb = [None if i is None else V[i] for i in self.b_idx]
        OW = [out[i] for i in self.w_idx]
        Ob = [None if i is None else out[i] for i in self.b_idx]
        self.plan.ggn_matvec(self.W, self.b, VW, Vb, OW, Ob, X, loss_kind, loss_scale, alpha, beta, aux=aux)

    def hessian_matvec(self, V: list[Tensor], out: list[Tensor], X: Tensor, G: Tensor, loss_kind: int,
                       loss_scale: float, alpha: float, beta: float) -> None:
        """``out = beta*out + alpha * H V`` (exact Hessian) for one mini-batch; ``G`` = gradient of the
        reduced mini-batch loss w.r.t. the model output."""
        VW = [V[i] for i in self.w_idx]
        Vb = [None if i is None else V[i] for i in self.b_idx]
        OW = [out[i] for i in self.w_idx]
        Ob = [None if i is None else out[i] for i in self.b_idx]
        self.plan.hessian_matvec(self.W, self.b, VW, Vb, OW, Ob, X, G, loss_kind, loss_scale, alpha, beta)

