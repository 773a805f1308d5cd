"""KFAC and EKFAC linear operators: ``P @ K @ P^T`` with ``K`` block-diagonal in the canonical
basis (one Kronecker / eigendecomposed block per parameter group).

Constructor signature, ``backend`` plug-in table, block layout ``Kronecker(G_l, A_l)`` (G first),
``inverse`` options and closed-form properties follow the reference
(``curvlinops/kfac.py:33-350``, ``ekfac.py:15-86``).  The ``"hip"`` backend (default) is the
MI355X-native computer of ``curvlinops_amd.computers``; the reference's name ``"hooks"`` is kept
as an alias so existing call sites run unchanged.
"""

from __future__ import annotations

from collections.abc import Callable, Iterable, MutableMapping

import torch
from torch import Tensor
from torch.nn import BCEWithLogitsLoss, CrossEntropyLoss, Module, MSELoss

from curvlinops_amd import linalg_native
from curvlinops_amd.canonical import FromCanonicalLinearOperator, ParamGroup, ToCanonicalLinearOperator
from curvlinops_amd.collector import CollectorEKFACComputer, CollectorKFACComputer
from curvlinops_amd.computers import HipEKFACComputer, HipKFACComputer
from curvlinops_amd.enums import FisherType, KFACType
from curvlinops_amd.kronecker import (
    BlockDiagonalLinearOperator,
    EighDecomposedLinearOperator,
    KroneckerProductLinearOperator,
)
from curvlinops_amd.linop import _ChainPyTorchLinearOperator


class KFACLinearOperator(_ChainPyTorchLinearOperator):
    """Kronecker-factored approximate curvature ``(A_l (x) G_l)`` per layer."""

    # "hip" / "hooks": module hooks (one use per parameter); "collector" / "make_fx": taps the affine
    # operations of the eager forward pass -- functional models and weight tying (reference's second
    # backend, `kfac.py:89-92`)
    _BACKENDS: dict[str, type] = {"hip": HipKFACComputer, "hooks": HipKFACComputer,
                                  "collector": CollectorKFACComputer, "make_fx": CollectorKFACComputer}
    SELF_ADJOINT: bool = True

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
        kfac_approx: str = KFACType.EXPAND,
        num_per_example_loss_terms: int | None = None,
        separate_weight_and_bias: bool = True,
        num_data: int | None = None,
        batch_size_fn: Callable[[MutableMapping | Tensor], int] | None = None,
        backend: str = "hip",
        distributed: bool = False,
    ):
        if backend not in self._BACKENDS:
            raise ValueError(f"Invalid backend: {backend!r}. Supported: {tuple(self._BACKENDS)}.")
        computer = self._BACKENDS[backend](
            model_func, loss_func, params, data, progressbar=progressbar,
            check_deterministic=check_deterministic, seed=seed, fisher_type=fisher_type,
            mc_samples=mc_samples, kfac_approx=kfac_approx,
            num_per_example_loss_terms=num_per_example_loss_terms,
            separate_weight_and_bias=separate_weight_and_bias, num_data=num_data,
            batch_size_fn=batch_size_fn, distributed=distributed,
        )
        K, mapping = self._compute_canonical_op(computer)
        P, PT = self._build_converters(computer, mapping)
        super().__init__(P, K, PT)
        self._distributed = distributed

    @staticmethod
    def _compute_canonical_op(computer) -> tuple[BlockDiagonalLinearOperator, list[ParamGroup]]:
        A, G, mapping = computer.compute()
        blocks = []
        for group in mapping:
            key = tuple(group.values())
            aaT, ggT = A.get(key), G[key]
            blocks.append(KroneckerProductLinearOperator(*([ggT, aaT] if aaT is not None else [ggT])))
        return BlockDiagonalLinearOperator(blocks), mapping

    @staticmethod
    def _build_converters(computer, mapping) -> tuple[FromCanonicalLinearOperator, ToCanonicalLinearOperator]:
        PT = ToCanonicalLinearOperator(
            {n: p.shape for n, p in computer._params.items()}, mapping, computer.device, computer.dtype
        )
        return PT.adjoint(), PT

    def trace(self) -> Tensor:
        return self[1].trace()

    def det(self) -> Tensor:
        return self[1].det()

    def logdet(self) -> Tensor:
        return self[1].logdet()

    def frobenius_norm(self) -> Tensor:
        return self[1].frobenius_norm()

    def inverse(
        self,
        damping: float = 0.0,
        use_heuristic_damping: bool = False,
        min_damping: float = 1e-8,
        use_exact_damping: bool = False,
        retry_double_precision: bool = True,
    ) -> _ChainPyTorchLinearOperator:
        P, K, PT = self
        if use_exact_damping and not use_heuristic_damping and all(
                isinstance(b, KroneckerProductLinearOperator) for b in K):
            # exact damping = eigendecompositions of all factors: decompose them together (equal sizes
            # batched, groups on worker threads) instead of one by one inside every block
            from curvlinops_amd.kronecker import ensure_all_square

            factors = [S for block in K for S in block]
            ensure_all_square(*factors)
            decs = linalg_native.eigh_many(factors)
            blocks, pos = [], 0
            for block in K:
                evals, evecs = zip(*decs[pos : pos + len(block)])
                pos += len(block)
                lam = evals[0]
                for e in evals[1:]:
                    lam = torch.kron(lam, e)
                blocks.append(EighDecomposedLinearOperator(lam, KroneckerProductLinearOperator(*evecs))
                              .inverse(damping=damping))
            return _ChainPyTorchLinearOperator(P, BlockDiagonalLinearOperator(blocks), PT)
        # the factors of all blocks are independent: their Cholesky inverses run concurrently (and,
        # for a data-parallel operator whose factors are replicated, sharded over the ranks)
        with linalg_native.concurrent_inverses(distributed=getattr(self, "_distributed", False)):
            K_inv = BlockDiagonalLinearOperator([
                block.inverse(
                    damping=damping, use_heuristic_damping=use_heuristic_damping, min_damping=min_damping,
                    use_exact_damping=use_exact_damping, retry_double_precision=retry_double_precision,
                )
                for block in K
            ])
        return _ChainPyTorchLinearOperator(P, K_inv, PT)


class EKFACLinearOperator(KFACLinearOperator):
    """Eigenvalue-corrected KFAC: ``(Q_g (x) Q_a) diag(lambda) (Q_g (x) Q_a)^T`` per layer."""

    _BACKENDS: dict[str, type] = {"hip": HipEKFACComputer, "hooks": HipEKFACComputer,
                                  "collector": CollectorEKFACComputer, "make_fx": CollectorEKFACComputer}

    @staticmethod
    def _compute_canonical_op(computer) -> tuple[BlockDiagonalLinearOperator, list[ParamGroup]]:
        Qa, Qg, lam, mapping = computer.compute()
        blocks = []
        for group in mapping:
            key = tuple(group.values())
            qa, qg = Qa.get(key), Qg[key]
            basis = KroneckerProductLinearOperator(*([qg, qa] if qa is not None else [qg]))
            blocks.append(EighDecomposedLinearOperator(lam[key].flatten(), basis))
        return BlockDiagonalLinearOperator(blocks), mapping

    def inverse(self, damping: float = 0.0) -> _ChainPyTorchLinearOperator:
        P, K, PT = self
        return _ChainPyTorchLinearOperator(
            P, BlockDiagonalLinearOperator([block.inverse(damping=damping) for block in K]), PT
        )
