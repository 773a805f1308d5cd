"""KFAC factors from the affine operations a forward pass actually executes (``backend="collector"``,
alias ``"make_fx"``).

The reference's second backend traces ``model_func`` with ``torch.fx`` / ``make_fx`` and pattern-matches
the graph for affine layers (``computers/io_collector/*``, ``computers/kfac_make_fx.py:26-111``): that is
what lets it handle *functional* models ``(params, X) -> prediction`` and **weight tying** (one parameter
used by several affine operations), where module hooks fire once per use with the wrong scaling
(``test/test_kfac.py:199-270``).  Here the same ``(a, g)`` contract is met without a tracing compiler:
a ``TorchFunctionMode`` taps every ``F.linear`` / ``F.conv2d`` call of the ordinary eager forward pass
whose weight or bias is one of the tracked parameter tensors and records (input, output, parameter
names, convolution geometry).  The uses of one weight are extra weight-sharing positions
(``io_collector/groups.py:117-160``: inputs concatenated along the shared axis): ``A = sum_u x_u^T x_u /
(N S_total)``, ``G = sum_u g_u^T g_u``; a use without the bias of a joint (W, b) group contributes a zero
instead of a one in the bias column.  Gradients w.r.t. the tapped outputs come from the same batched
backward pass as the hooks backend, the Gram matrices from the same HIP SYRK kernels.
"""

from __future__ import annotations

from functools import partial

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.overrides import TorchFunctionMode

from curvlinops_amd.canonical import ParamGroup
from curvlinops_amd.computers import (
    HipKFACComputer, _FactorStore, _gram_accumulate, _join_factor_stream, input_to_weight_sharing_format,
)
from curvlinops_amd.enums import FisherType
from curvlinops_amd.utils import seed_generator


class _Use:
    """One affine operation of the forward pass."""

    __slots__ = ("x", "out", "W", "b", "hyper")

    def __init__(self, x, out, W, b, hyper):
        self.x, self.out, self.W, self.b, self.hyper = x, out, W, b, hyper


class _AffineTap(TorchFunctionMode):
    def __init__(self, names_by_id: dict[int, str]):
        super().__init__()
        self._names = names_by_id
        self.uses: list[_Use] = []

    def _name(self, t) -> str | None:
        if not isinstance(t, Tensor):
            return None
        hit = self._names.get(id(t))
        if hit is None and t._base is not None and id(t._base) in self._names:
            # `W.T`, `W.view(...)`, slices: the factor contract is defined for the parameter itself
            # (`layer_io.py:128-336` matches parameters, not their views) -- dropping such a use silently
            # would give wrong factors under weight tying
            raise NotImplementedError(
                f"collector backend: a view of tracked parameter {self._names[id(t._base)]!r} is used as the weight / "
                "bias of an affine operation; pass the parameter itself (views of tied weights are not supported)")
        return hit

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        if func is F.linear:
            x = args[0] if args else kwargs["input"]
            W = args[1] if len(args) > 1 else kwargs["weight"]
            b = args[2] if len(args) > 2 else kwargs.get("bias")
            wn, bn = self._name(W), self._name(b)
            if wn is not None or bn is not None:
                self.uses.append(_Use(x, out, wn, bn, {}))
        elif func is F.conv2d:
            names = ("input", "weight", "bias", "stride", "padding", "dilation", "groups")
            defaults = {"bias": None, "stride": 1, "padding": 0, "dilation": 1, "groups": 1}
            bound = {**defaults, **dict(zip(names, args)), **kwargs}
            wn, bn = self._name(bound["weight"]), self._name(bound["bias"])
            if wn is not None or bn is not None:
                W = bound["weight"]
                hyper = dict(kernel_size=tuple(W.shape[2:]), stride=bound["stride"], padding=bound["padding"],
                             dilation=bound["dilation"], groups=bound["groups"])
                self.uses.append(_Use(bound["input"], out, wn, bn, hyper))
        return out


class CollectorKFACComputer(HipKFACComputer):
    """KFAC's Kronecker factors for nn.Modules AND functional models, exact under weight tying."""

    _REQUIRES_MODULE = False

    def compute(self):
        return self._compute_kronecker_factors()

    # ------------------------------------------------------------------ parameter groups from the uses
    def _groups_from_uses(self, uses: list[_Use]) -> list[ParamGroup]:
        modules: dict[str, ParamGroup] = {}
        for u in uses:
            if u.W is not None:
                mod = modules.setdefault(u.W, {"W": u.W})
                if u.b is not None:
                    if not self._separate_weight_and_bias and mod.get("b") not in (None, u.b):
                        raise ValueError(
                            f"Weight '{u.W}' is used with conflicting biases '{mod['b']}' and '{u.b}' under joint "
                            "treatment. Use separate_weight_and_bias=True."
                        )
                    mod["b"] = u.b
            elif u.b is not None:
                modules.setdefault(u.b, {"b": u.b})
        groups: list[ParamGroup] = []
        for mod in modules.values():
            groups.extend([{r: n} for r, n in mod.items()] if self._separate_weight_and_bias else [mod])
        seen = {n for g in groups for n in g.values()}
        missing = set(self._params) - seen
        if missing:
            raise NotImplementedError(
                f"Parameters {missing} are not used as the weight / bias of a linear or 2d-convolution operation."
            )
        return groups

    # ------------------------------------------------------------------ the sweep over the data
    def _compute_kronecker_factors(self):
        A, G = _FactorStore(), _FactorStore()
        params = {n: (p if p.requires_grad else p.detach().requires_grad_(True)) for n, p in self._params.items()}
        ids = {id(t): n for n, t in params.items()}
        mapping: list[ParamGroup] | None = None
        self._generator = seed_generator(self._generator, self.device, self._seed)
        self._hooked_outputs = []
        try:
            for X, y in self._loop_over_data(desc="KFAC matrices"):
                with _AffineTap(ids) as tap:
                    output = self._model_func(params, X)
                uses = tap.uses
                groups = self._groups_from_uses(uses)
                if mapping is None:
                    mapping = groups
                elif groups != mapping:
                    raise RuntimeError("The affine operations of the model changed between mini-batches.")
                for group in mapping:
                    # the uses of the group's weight; a bias-only group: the uses that add this bias
                    mine = [u for u in uses if (u.W == group["W"] if "W" in group else u.b == group["b"])]
                    for u in mine:
                        self._track_output(u.out, partial(self._grad_hook, group=group, hyper=u.hyper, store=G))
                    if "W" in group:
                        self._accumulate_inputs(A, group, mine)
                output, y = self._rearrange_output(output, y)
                self._backpropagate(output, y)
        finally:
            self._hooked_outputs = []
            _join_factor_stream(self.device)
        if mapping is None:
            raise ValueError("The data iterable is empty.")
        if self._distributed:
            from curvlinops_amd.dist import allreduce_tensors_

            allreduce_tensors_([*A.values(), *G.values()])
        if self._fisher_type == FisherType.FORWARD_ONLY:
            for group in mapping:
                p = self._params[next(iter(group.values()))]
                G[tuple(group.values())] = torch.eye(p.shape[0], dtype=p.dtype, device=self.device)
        return dict(A), dict(G), mapping

    def _accumulate_inputs(self, A: dict, group: ParamGroup, uses: list[_Use]) -> None:
        """``A += sum_u [x_u | bias column]^T [x_u | bias column] / (N_data * S_total)``."""
        joint = "b" in group
        xs = [input_to_weight_sharing_format(u.x.data.detach(), self._kfac_approx, u.hyper) for u in uses]
        shared = sum(x.shape[1] for x in xs)
        key = tuple(group.values())
        for u, x in zip(uses, xs):
            x2d = x.reshape(-1, x.shape[-1])
            with_bias = joint and u.b is not None
            if joint and not with_bias:  # this use does not add the group's bias: zero bias column
                x2d = torch.cat([x2d, x2d.new_zeros(x2d.shape[0], 1)], dim=1)
            _gram_accumulate(A, key, x2d, 1.0 / (self._N_data * shared), ones_col=with_bias)


class CollectorEKFACComputer(CollectorKFACComputer):
    """EKFAC for functional models / weight tying (reference ``computers/ekfac_make_fx.py``): KFAC factors
    from the taps above -> eigenbases -> eigenvalues re-fitted in that basis with the per-example gradients of
    a group assembled over ALL uses of its weight (inputs and output-gradients concatenated along the shared
    axis, ``io_collector/groups.py:117-160``)."""

    _SUPPORTED_FISHER_TYPE = (FisherType.TYPE2, FisherType.MC, FisherType.EMPIRICAL)

    def _rearrange_output(self, output: Tensor, y: Tensor) -> tuple[Tensor, Tensor]:
        if output.ndim != 2 or y.ndim not in {1, 2}:
            raise ValueError(
                f"Only 2d output and 1d/2d target are supported for EKFAC. Got {output.ndim=} and {y.ndim=}."
            )
        return output, y

    def compute(self):
        from curvlinops_amd import linalg_native
        from curvlinops_amd.computers import (
            compute_eigenvalue_correction, compute_loss_correction, grad_to_weight_sharing_format,
        )
        from curvlinops_amd.enums import KFACType

        A, G, mapping = self._compute_kronecker_factors()
        keys = [("a", k) for k in A] + [("g", k) for k in G]
        bases = linalg_native.eigh_many([A[k] if w == "a" else G[k] for w, k in keys])
        Qa = {k: q[1] for (w, k), q in zip(keys, bases) if w == "a"}
        Qg = {k: q[1] for (w, k), q in zip(keys, bases) if w == "g"}
        lam: dict = {}
        params = {n: (p if p.requires_grad else p.detach().requires_grad_(True)) for n, p in self._params.items()}
        ids = {id(t): n for n, t in params.items()}
        self._generator = seed_generator(self._generator, self.device, self._seed)
        for X, y in self._loop_over_data(desc="Eigenvalue correction"):
            with _AffineTap(ids) as tap:
                output = self._model_func(params, X)
            uses = tap.uses
            output, y = self._rearrange_output(output, y)
            grad_outputs = self._grad_outputs_computer(output.detach(), y, self._generator)  # [V, B, C]
            if self._loss_func.reduction == "mean":
                grad_outputs.mul_(1.0 / output.shape[0])
            outs = [u.out for u in uses]
            V = grad_outputs.shape[0]
            per_v = [torch.autograd.grad(output, outs, grad_outputs=grad_outputs[v], retain_graph=v < V - 1,
                                         allow_unused=True) for v in range(V)]
            batch_size = output.shape[0]
            corr = compute_loss_correction(batch_size, self._num_per_example_loss_terms,
                                           self._loss_func.reduction, self._N_data)
            for group in mapping:
                key = tuple(group.values())
                idx = [i for i, u in enumerate(uses) if (u.W == group["W"] if "W" in group else u.b == group["b"])]
                # [V, B, S_total, d_out]
                g = torch.cat([torch.stack([grad_to_weight_sharing_format(per_v[v][i].detach(), KFACType.EXPAND,
                                                                          uses[i].hyper) for v in range(V)])
                               for i in idx], dim=2)
                a = None
                if "W" in group:
                    cols = []
                    for i in idx:
                        x = input_to_weight_sharing_format(uses[i].x.data.detach(), KFACType.EXPAND, uses[i].hyper)
                        if "b" in group:
                            fill = x.new_ones if uses[i].b is not None else x.new_zeros
                            x = torch.cat([x, fill(*x.shape[:-1], 1)], dim=-1)
                        cols.append(x)
                    a = torch.cat(cols, dim=1)
                upd = compute_eigenvalue_correction(g, Qg[key], a, Qa.get(key)).mul_(corr)
                if key in lam:
                    lam[key].add_(upd)
                else:
                    lam[key] = upd
        return Qa, Qg, lam, mapping
