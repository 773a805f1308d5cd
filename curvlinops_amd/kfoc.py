"""KFOC: the Frobenius-optimal rank-one Kronecker approximation of each layer's GGN block
(reference ``curvlinops/kfoc.py:13-106``, ``computers/kfoc_make_fx.py:31-271``; Schnaus et al. 2021,
Koroko et al. 2022).

``S_1, S_2 = argmin || G_l - S_1 (x) S_2 ||_F`` is the top singular pair of the Van Loan
rearrangement ``R(G_l)``, a ``d_out^2 x d_in^2`` operator that acts on matrices as

    R(G) vec(M)   = vec( sum_{v,n} P_vn M P_vn^T ),      R(G)^T vec(U) = vec( sum_{v,n} P_vn^T U P_vn ),

with the per-sample weight gradients ``P_vn = sum_s g_vns a_ns^T``.

Design differences to the reference (same results):

* The reference materialises ``P`` (``V B d_out d_in`` floats per layer) and contracts it with a
  three-operand einsum per product.  Here ``P_vn = G_vn^T A_n`` stays FACTORED (``A_n`` = the layer
  inputs ``[S, d_in]``, ``G_vn`` = the output gradients ``[S, d_out]``):
  ``sum P M P^T = sum G_vn^T (A_n M A_n^T) G_vn`` is two large GEMMs and two small batched ones
  (for layers without weight sharing, ``S = 1``, the middle factor is a scalar per sample), all on
  the fp32 MFMA GEMM engine for GPU tensors.
* The reference hands the operator to SciPy's ARPACK ``svds`` on the host (one PCIe round trip
  per product).  Here the top singular triplet comes from a device-resident Golub-Kahan-Lanczos
  bidiagonalisation with full reorthogonalisation, started at ``vec(I)`` (``R(G) vec(I)`` is KFAC's
  gradient factor up to scale, so the start is already close); only the small bidiagonal matrix goes
  to the host.  The pair is sign-normalised so that ``trace(S_2) >= 0`` (the product is unaffected).
* Layer inputs / output gradients are collected by the module hooks of the ``"hip"`` KFAC computer
  (one forward pass, ONE batched backward pass for the V backpropagated vectors) instead of an FX trace.
"""

from __future__ import annotations

from collections.abc import Callable, Iterable, MutableMapping
from functools import partial
from math import sqrt

import numpy as np
import torch
from torch import Tensor
from torch.nn import BCEWithLogitsLoss, CrossEntropyLoss, Module, MSELoss

from curvlinops_amd import _hip
from curvlinops_amd.computers import (
    HipKFACComputer,
    ParamGroupKey,
    _conv_hyperparams,
    _use_params,
    grad_to_weight_sharing_format,
    input_to_weight_sharing_format,
    seed_generator,
)
from curvlinops_amd.enums import FisherType, KFACType
from curvlinops_amd.kfac import KFACLinearOperator
from curvlinops_amd.linop import PyTorchLinearOperator
from curvlinops_amd.utils import is_native_tensor


def _mm(A: Tensor, B: Tensor) -> Tensor:
    """2-D or batched product: the MFMA GEMM engine for fp32 GPU tensors, torch otherwise."""
    if is_native_tensor(A) and is_native_tensor(B) and A.numel() and B.numel():
        return _hip.gemm(A, B)
    return A @ B


def rearranged_ggn_apply(a: Tensor, g: Tensor, M: Tensor, adjoint: bool = False) -> Tensor:
    """``sum_{v,n} P_vn M P_vn^T`` (or ``sum P_vn^T M P_vn`` for ``adjoint``) with
    ``P_vn = g[v, n]^T a[n]`` kept factored.  ``a``: ``[B, S, d_in]``, ``g``: ``[V, B, S, d_out]``."""
    B, S, d_in = a.shape
    V, _, _, d_out = g.shape
    if not adjoint:
        T = _mm(a.reshape(B * S, d_in), M).reshape(B, S, d_in)
        if S == 1:
            q = (T * a).sum(-1)  # [B, 1]: a_n^T M a_n
            W = g * q[None, :, :, None]
        else:
            C = _mm(T, a.transpose(1, 2))  # [B, S, S] = A_n M A_n^T
            W = torch.stack([_mm(C, g[v]) for v in range(V)])
        return _mm(g.reshape(-1, d_out).T, W.reshape(-1, d_out))
    T = _mm(g.reshape(-1, d_out), M).reshape(V, B, S, d_out)
    if S == 1:
        q = (T * g).sum(dim=(0, 3))  # [B, 1]: sum_v g_vn^T U g_vn
        W = a * q[:, :, None]
    else:
        C = _mm(T[0], g[0].transpose(1, 2))
        for v in range(1, V):
            C = C + _mm(T[v], g[v].transpose(1, 2))
        W = _mm(C, a)
    return _mm(a.reshape(-1, d_in).T, W.reshape(-1, d_in))


def rearranged_ggn_apply_explicit(P: Tensor, M: Tensor, adjoint: bool = False) -> Tensor:
    """The same products from materialised per-sample gradients ``P``: ``[R, d_out, d_in]`` (``R`` = all
    (vector, sample) pairs).  Cheaper than the factored form when a layer has many shared positions
    but few weights per sample pair (early convolutions: ``S = 784``, ``d_out d_in = 6 x 26``)."""
    R, d_out, d_in = P.shape
    if not adjoint:
        T = _mm(P.reshape(R * d_out, d_in), M).reshape(R, d_out, d_in)
        return _mm(T, P.transpose(1, 2)).sum(0)
    T = _mm(P.transpose(1, 2), M)  # [R, d_in, d_out]
    return _mm(T, P).sum(0)


def _explicit_is_cheaper(a: Tensor, g: Tensor) -> bool:
    """Per-sample work of the two forms: ``V d_out d_in (d_in + d_out)`` against
    ``S^2 (d_in + V d_out)`` (the ``S x S`` inner matrices of the factored form)."""
    S, d_in = a.shape[1:]
    V, d_out = g.shape[0], g.shape[-1]
    return S > 1 and V * d_out * d_in * (d_in + d_out) < S * S * (d_in + V * d_out)


class _RearrangedGGNLinearOperator(PyTorchLinearOperator):
    r"""Van Loan rearrangement of a per-layer GGN block, :math:`d_\text{out}^2 \times d_\text{in}^2`
    (reference ``computers/kfoc_make_fx.py:31-120``).  Built either from explicit per-sample
    gradients ``[V, N, d_out, d_in]`` (the reference's constructor) or, through :meth:`from_io`,
    from the factored layer inputs / output gradients."""

    def __init__(self, per_sample_grads: Tensor, adjoint: bool = False):
        V, N, d_out, d_in = per_sample_grads.shape
        self._P = per_sample_grads.reshape(V * N, d_out, d_in)
        self._setup(None, None, adjoint, d_in, d_out)

    @classmethod
    def from_io(cls, a: Tensor, g: Tensor, adjoint: bool = False) -> "_RearrangedGGNLinearOperator":
        self = cls.__new__(cls)
        self._setup(a, g, adjoint)
        return self

    def _setup(self, a: Tensor | None, g: Tensor | None, adjoint: bool, d_in: int | None = None,
               d_out: int | None = None) -> None:
        if a is not None:
            d_in, d_out = a.shape[-1], g.shape[-1]
            self._P = None
        in_shape = [(d_out, d_out)] if adjoint else [(d_in, d_in)]
        out_shape = [(d_in, d_in)] if adjoint else [(d_out, d_out)]
        PyTorchLinearOperator.__init__(self, in_shape, out_shape)
        self._a, self._g, self._is_adjoint = a, g, adjoint

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        (M,) = X
        cols = [self._apply(M[..., k].contiguous()) for k in range(M.shape[-1])]
        return [torch.stack(cols, dim=-1)]

    def _apply(self, M: Tensor) -> Tensor:
        if self._P is not None:
            return rearranged_ggn_apply_explicit(self._P, M, self._is_adjoint)
        return rearranged_ggn_apply(self._a, self._g, M, self._is_adjoint)

    def _adjoint(self) -> "_RearrangedGGNLinearOperator":
        if self._P is not None:
            return type(self)(self._P.unsqueeze(0), adjoint=not self._is_adjoint)
        return type(self).from_io(self._a, self._g, adjoint=not self._is_adjoint)

    @property
    def device(self) -> torch.device:
        return (self._P if self._P is not None else self._a).device

    @property
    def dtype(self) -> torch.dtype:
        return (self._P if self._P is not None else self._a).dtype


def top_singular_triplet(apply: Callable[[Tensor], Tensor], apply_t: Callable[[Tensor], Tensor], v0: Tensor,
                         tol: float, max_steps: int = 48) -> tuple[float, Tensor | None, Tensor | None]:
    """Largest singular value and vectors of the linear map ``apply`` (adjoint ``apply_t``) by
    Golub-Kahan-Lanczos bidiagonalisation with full reorthogonalisation, started at ``v0``.
    Operands keep their (matrix) shapes.  The Lanczos bases live in two device buffers
    (reorthogonalisation = two GEMVs per side); the host only sees the bidiagonal coefficients
    ``R V_k = U_k B_k`` and stops on the residual estimate ``beta_k |x_k| <= tol * sigma``.
    Returns ``(0, None, None)`` for the zero map."""
    limit = max(1, min(max_steps, v0.numel()))
    out_shape = None
    Vb = v0.new_zeros(limit + 1, v0.numel())
    Ub: Tensor | None = None
    Vb[0] = v0.reshape(-1) / torch.linalg.vector_norm(v0)
    alphas: list[float] = []
    betas: list[float] = []
    sigma, x, y = 0.0, None, None

    def reorth(vec: Tensor, basis: Tensor) -> Tensor:
        for _ in range(2):  # classical Gram-Schmidt, twice
            vec = vec - basis.T @ (basis @ vec)
        return vec

    for j in range(limit):
        u = apply(Vb[j].reshape(v0.shape))
        if Ub is None:
            out_shape = u.shape
            Ub = u.new_zeros(limit, u.numel())
        u = u.reshape(-1)
        if j > 0:
            u = reorth(u - betas[j - 1] * Ub[j - 1], Ub[:j])
        alpha = float(torch.linalg.vector_norm(u))
        if alpha == 0.0 or (alphas and alpha <= 1e-13 * alphas[0]):
            if not alphas:
                return 0.0, None, None
            # the left Krylov space is exhausted (e.g. an exactly Kronecker-structured block):
            # R V_{k+1} = U_k B with the k x (k+1) bidiagonal B holds exactly -> exact triplet
            k = len(alphas)
            Bk = np.zeros((k, k + 1))
            Bk[np.arange(k), np.arange(k)] = alphas
            Bk[np.arange(k), np.arange(1, k + 1)] = betas
            X, sv, Yt = np.linalg.svd(Bk, full_matrices=False)
            sigma, x, y = float(sv[0]), X[:, 0], Yt[0]
            break
        Ub[j] = u / alpha
        alphas.append(alpha)
        w = apply_t(Ub[j].reshape(out_shape)).reshape(-1) - alpha * Vb[j]
        w = reorth(w, Vb[: j + 1])
        beta = float(torch.linalg.vector_norm(w))
        k = len(alphas)
        Bk = np.diag(np.array(alphas)) + (np.diag(np.array(betas), 1) if k > 1 else 0.0)
        X, sv, Yt = np.linalg.svd(Bk)
        sigma, x, y = float(sv[0]), X[:, 0], Yt[0]
        if beta * abs(x[-1]) <= tol * sigma or beta <= 1e-13 * alphas[0] or k == limit:
            break
        betas.append(beta)
        Vb[j + 1] = w / beta
    u = (torch.as_tensor(x, dtype=Ub.dtype, device=Ub.device) @ Ub[: len(x)]).reshape(out_shape)
    v = (torch.as_tensor(y, dtype=Vb.dtype, device=Vb.device) @ Vb[: len(y)]).reshape(v0.shape)
    return sigma, u, v


def top_rank_one_kron_factors(a: Tensor, g: Tensor) -> tuple[Tensor, Tensor]:
    """``(S_1 [d_out, d_out], S_2 [d_in, d_in])`` with ``S_1 (x) S_2`` the Frobenius-optimal rank-one
    Kronecker approximation of ``G = sum_{v,n} vec(P_vn) vec(P_vn)^T`` (reference
    ``computers/kfoc_make_fx.py:123-180``); zero factors for ``G = 0``."""
    d_in, d_out = a.shape[-1], g.shape[-1]
    eye = torch.eye(d_in, dtype=a.dtype, device=a.device)
    tol = 1e-11 if a.dtype == torch.float64 else 1e-6
    if _explicit_is_cheaper(a, g):
        V, B, S, _ = g.shape
        P = _mm(g.reshape(V * B, S, d_out).transpose(1, 2), a.repeat(V, 1, 1) if V > 1 else a)  # [V B, d_out, d_in]
        fwd, bwd = partial(rearranged_ggn_apply_explicit, P), partial(rearranged_ggn_apply_explicit, P, adjoint=True)
    else:
        fwd, bwd = partial(rearranged_ggn_apply, a, g), partial(rearranged_ggn_apply, a, g, adjoint=True)
    sigma, u, v = top_singular_triplet(fwd, bwd, eye, tol)
    if u is None:
        return a.new_zeros(d_out, d_out), a.new_zeros(d_in, d_in)
    sign = -1.0 if float(torch.trace(v)) < 0 else 1.0
    scale = sqrt(sigma)
    return (sign * scale) * u, (sign * scale) * v


class HipKFOCComputer(HipKFACComputer):
    """KFOC factors from one forward + one batched backward pass over a SINGLE batch."""

    _SUPPORTED_FISHER_TYPE: tuple[FisherType, ...] = (FisherType.TYPE2, FisherType.MC)
    _SUPPORTED_KFAC_APPROX: tuple[KFACType, ...] = (KFACType.EXPAND,)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if len(list(self._data)) != 1:
            raise ValueError("KFOC supports a single batch of data only.")

    def compute(self):
        with _use_params(self._model_module, self._params):
            return self._compute_kfoc_factors()

    def _compute_kfoc_factors(self):
        mapping = self.compute_parameter_groups(self._params, self._model_module, self._separate_weight_and_bias)
        a_of: dict[ParamGroupKey, Tensor] = {}
        g_of: dict[ParamGroupKey, Tensor] = {}
        handles = []
        for group in mapping:
            mod = self._module_of(group)
            hyper = _conv_hyperparams(mod)
            key = tuple(group.values())
            if "W" in group:
                handles.append(mod.register_forward_pre_hook(partial(self._store_input, key=key, group=group, hyper=hyper, store=a_of)))
            handles.append(mod.register_forward_hook(partial(self._store_grad_later, key=key, hyper=hyper, store=g_of)))
        self._generator = seed_generator(self._generator, self.device, self._seed)
        self._hooked_outputs = []
        try:
            X, y = next(iter(self._loop_over_data(desc="KFOC")))
            output = self._model_module(X)
            output, y = self._rearrange_output(output, y)
            n_terms = output.shape[0]
            self._backpropagate(output, y)
        finally:
            for h in handles:
                h.remove()
        # _backpropagate scales the vectors by 1/N for mean reductions (the hooks convention);
        # here sum_v g g^T must be the batch loss Hessian itself: 1/sqrt(N) per vector
        # (reference io_collector/layer_io.py:177-182)
        fix = sqrt(n_terms) if self._loss_func.reduction == "mean" else 1.0
        first: dict[ParamGroupKey, Tensor] = {}
        second: dict[ParamGroupKey, Tensor] = {}
        for group in mapping:
            key = tuple(group.values())
            g = g_of[key] * fix  # [V, B, S, d_out]
            if "W" in group:
                first[key], second[key] = top_rank_one_kron_factors(a_of[key], g)
            else:  # bias-only block: the exact GGN block is the optimum
                b = g.sum(dim=2).reshape(-1, g.shape[-1])
                first[key] = _mm(b.T.contiguous(), b)
        return second, first, mapping

    @property
    def _manual_callbacks(self) -> bool:
        return True  # always one batched backward; the callbacks see [V, B, ...]

    def _store_input(self, module, inputs, key, group, hyper, store) -> None:
        if len(inputs) != 1:
            raise ValueError("Modules with multiple inputs are not supported.")
        x = input_to_weight_sharing_format(inputs[0].data.detach(), self._kfac_approx, hyper)
        if "b" in group:  # joint weight + bias: a column of ones
            x = torch.cat([x, x.new_ones(*x.shape[:-1], 1)], dim=-1)
        store[key] = x.contiguous()

    def _store_grad_later(self, module, inputs, output, key, hyper, store) -> None:
        def cb(grad: Tensor, stacked: bool = False) -> None:
            g = grad.data.detach()
            if not stacked:  # sequential fallback: one call per backpropagated vector
                g = g.unsqueeze(0)
            V, B = g.shape[:2]
            gs = grad_to_weight_sharing_format(g.flatten(0, 1), self._kfac_approx, hyper)  # [V B, S, d_out]
            gs = gs.reshape(V, B, *gs.shape[1:])
            store[key] = torch.cat([store[key], gs]) if key in store else gs.contiguous()

        self._track_output(output, cb)


class KFOCLinearOperator(KFACLinearOperator):
    r"""Frobenius-optimal rank-one Kronecker approximation of the GGN, block-diagonal over layers.

    Scope as in the reference: a single batch of data, ``fisher_type`` in ``{TYPE2, MC}``.
    Symmetry / positive semi-definiteness of the factors is not enforced."""

    _BACKENDS: dict[str, type] = {"hip": HipKFOCComputer, "make_fx": HipKFOCComputer}

    def __init__(
        self,
        model_func: Module | Callable[[dict[str, Tensor], Tensor | MutableMapping], Tensor],
        loss_func: MSELoss | CrossEntropyLoss | BCEWithLogitsLoss,
        params: dict[str, Tensor],
        data: Iterable[tuple[Tensor | MutableMapping, Tensor]],
        progressbar: bool = False,
        check_deterministic: bool = True,
        seed: int = 2_147_483_647,
        fisher_type: str = FisherType.MC,
        mc_samples: int = 1,
        separate_weight_and_bias: bool = True,
        num_data: int | None = None,
        batch_size_fn: Callable[[MutableMapping | Tensor], int] | None = None,
    ):
        super().__init__(
            model_func, loss_func, params, data, progressbar=progressbar,
            check_deterministic=check_deterministic, seed=seed, fisher_type=fisher_type,
            mc_samples=mc_samples, kfac_approx=KFACType.EXPAND,
            separate_weight_and_bias=separate_weight_and_bias, num_data=num_data,
            batch_size_fn=batch_size_fn, backend="hip",
        )


__all__ = ["KFOCLinearOperator", "HipKFOCComputer", "rearranged_ggn_apply", "rearranged_ggn_apply_explicit",
           "top_rank_one_kron_factors",
           "top_singular_triplet"]
