"""Kronecker-product, eigendecomposed and block-diagonal operators (the canonical-space
building blocks of KFAC / EKFAC).

Semantics follow the reference (``curvlinops/kronecker.py:47-373``, ``eigh.py:14-177``,
``blockdiagonal.py:18-189``): ``(S_1 (x) S_2 (x) ...) x`` on a flat vector with K trailing;
closed-form trace / det / logdet / Frobenius norm; ``inverse`` with plain, heuristic
(Martens & Grosse 2015, section 6.3) or exact damping; fp64 retry when Cholesky fails.

On fp32 GPU tensors every contraction runs on the f32-MFMA GEMM of ``libclo_hip`` (the
``einsum('abZ,Aa,Bb->ABZ')`` of ``kronecker.py:85,153`` becomes two GEMMs on a K-major copy of
the operand); other dtypes / devices use ``torch.einsum``.
"""

from __future__ import annotations

from collections.abc import Iterator
from math import prod, sqrt

import torch
from torch import Tensor

from curvlinops_amd import _hip, linalg_native
from curvlinops_amd.linop import (
    PyTorchLinearOperator,
    _expect_same_device,
    _expect_same_dtype,
    _expect_same_shape,
    _expect_same_spaces,
)
from curvlinops_amd.canonical import is_kmajor
from curvlinops_amd.utils import infer_device, infer_dtype, is_native_tensor, side_stream, split_list


def ensure_all_square(*objs) -> None:
    for o in objs:
        if len(o.shape) != 2 or o.shape[0] != o.shape[1]:
            raise RuntimeError(f"{type(o)} is not square: {o.shape}.")


def _row_major(S: Tensor) -> bool:
    return S.dim() == 2 and (S.shape[1] == 1 or S.stride(1) == 1) and (S.shape[0] == 1 or S.stride(0) >= S.shape[1])


def _contiguous_pair(factors, transpose: bool):
    """``(S1, S2, flags)`` with row-major arrays (any leading dimension) for ``clo_kron_matmat``: bit i-1 of ``flags`` says
    that array i holds the TRANSPOSE of the factor the product needs -- adjoint operators and the eigensolver's
    row-eigenvector arrays arrive as ``.T`` views of row-major arrays; None if a factor is neither."""
    arrays, flags = [], 0
    for i, S in enumerate(factors):
        if _row_major(S):
            arrays.append(S)
            flags |= (1 << i) if transpose else 0
        elif _row_major(S.T):
            arrays.append(S.T)
            flags |= 0 if transpose else (1 << i)
        else:
            return None
    return arrays[0], arrays[1], flags


def _kron_apply_native(factors: list[Tensor], x: Tensor, transpose: bool) -> Tensor:
    """``(S_1 (x) ... (x) S_n) x`` for ``x [prod(in dims), K]`` on the HIP GEMM."""
    K = x.shape[-1]
    ins = [S.shape[0] if transpose else S.shape[1] for S in factors]
    mats = [S.T if transpose else S for S in factors]
    if len(factors) == 2 and (K == 1 or is_kmajor(x)):
        # the vector, or a K-major operand (from ToCanonical's fused pack or from another block): ONE foreign call, no
        # transposes, and the result stays K-major for the consumer
        fc = _contiguous_pair(factors, transpose)
        if fc is not None:
            xk = x.T if K > 1 else x.reshape(1, -1)
            if xk.is_contiguous():
                y = _hip.kron_matmat(fc[0], fc[1], xk, K, trans=fc[2])
                return y.T if K > 1 else y.reshape(-1, 1)
    if len(factors) == 2 and is_kmajor(x):
        S1, S2 = mats
        a, b = ins
        A_, B_ = S1.shape[0], S2.shape[0]
        T = _hip.gemm(x.T.view(K * a, b), S2.T)          # [(K a), B]
        Y = _hip.gemm(S1, T.view(K, a, B_))              # [K, A, B]
        return Y.view(K, A_ * B_).T
    xc = x.contiguous()
    if len(factors) == 1:
        return _hip.gemm(mats[0], xc)
    if len(factors) == 2:
        S1, S2 = mats
        a, b = ins
        A_, B_ = S1.shape[0], S2.shape[0]
        if K == 1:
            T = _hip.gemm(S1, xc.view(a, b))             # [A, b]
            return _hip.gemm(T, S2.T).view(A_ * B_, 1)   # [A, B]
        xk = _hip.transpose(xc.view(a * b, K))           # [K, a*b]  (K-major)
        T = _hip.gemm(xk.view(K * a, b), S2.T)           # [(K a), B]
        Y = _hip.gemm(S1, T.view(K, a, B_))              # batched over K: [K, A, B]
        return _hip.transpose(Y.view(K, A_ * B_))        # [A*B, K]
    # general case: contract one mode at a time
    cur = xc.view(*ins, K)
    for i, S in enumerate(mats):
        moved = cur.movedim(i, 0).contiguous()
        rest = moved.shape[1:]
        res = _hip.gemm(S, moved.view(moved.shape[0], -1))
        cur = res.view(S.shape[0], *rest).movedim(0, i)
    return cur.reshape(-1, K)


class KroneckerProductLinearOperator(PyTorchLinearOperator):
    """``S_1 (x) S_2 (x) ...`` acting on flattened tensors (one flat input/output space)."""

    def __init__(self, *factors: Tensor):
        if len(factors) == 0:
            raise ValueError("At least one factor must be provided.")
        for i, f in enumerate(factors):
            if f.ndim != 2:
                raise ValueError(f"Factor {i} must be a 2D tensor, got shape {f.shape}.")
        if len(factors) > 25:
            raise ValueError(f"At most 25 Kronecker factors supported, got {len(factors)}")
        self._factors = list(factors)
        d_in = prod(S.shape[1] for S in factors)
        d_out = prod(S.shape[0] for S in factors)
        super().__init__([(d_in,)], [(d_out,)])

    # container protocol -------------------------------------------------------------
    def __iter__(self) -> Iterator[Tensor]:
        return iter(self._factors)

    def __len__(self) -> int:
        return len(self._factors)

    def __getitem__(self, index: int) -> Tensor:
        return self._factors[index]

    def __setitem__(self, index: int, value: Tensor) -> None:
        old = self._factors[index]
        _expect_same_shape(old, value)
        _expect_same_device(old, value)
        _expect_same_dtype(old, value)
        self._factors[index] = value

    # products -------------------------------------------------------------------------
    def _apply(self, x: Tensor, transpose: bool) -> Tensor:
        fs = self._factors
        if is_native_tensor(x) and all(is_native_tensor(S) for S in fs):
            return _kron_apply_native(fs, x, transpose)
        n = len(fs)
        ins = [S.shape[0] if transpose else S.shape[1] for S in fs]
        lo = [chr(ord("a") + i) for i in range(n)]
        up = [chr(ord("A") + i) for i in range(n)]
        subs = ",".join(f"{u}{l}" for u, l in zip(up, lo))
        if transpose:
            eq = f"{''.join(up)}Z,{subs}->{''.join(lo)}Z"
        else:
            eq = f"{''.join(lo)}Z,{subs}->{''.join(up)}Z"
        return torch.einsum(eq, x.reshape(*ins, x.shape[-1]), *fs).flatten(end_dim=-2)

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        (x,) = X
        return [self._apply(x, transpose=False)]

    def _adjoint_matmat(self, X: list[Tensor]) -> list[Tensor]:
        (x,) = X
        return [self._apply(x, transpose=True)]

    def _adjoint(self) -> "KroneckerProductLinearOperator":
        return KroneckerProductLinearOperator(*[S.adjoint() for S in self._factors])

    @property
    def device(self) -> torch.device:
        return infer_device(self._factors)

    @property
    def dtype(self) -> torch.dtype:
        return infer_dtype(self._factors)

    # closed-form properties --------------------------------------------------------------
    def trace(self) -> Tensor:
        ensure_all_square(*self._factors)
        return torch.stack([S.trace() for S in self._factors]).prod()

    def det(self) -> Tensor:
        ensure_all_square(*self._factors)
        dim = prod(S.shape[0] for S in self._factors)
        return torch.stack([S.det() ** (dim // S.shape[0]) for S in self._factors]).prod()

    def logdet(self) -> Tensor:
        ensure_all_square(*self._factors)
        dim = prod(S.shape[0] for S in self._factors)
        return torch.stack([(dim // S.shape[0]) * S.logdet() for S in self._factors]).sum()

    def frobenius_norm(self) -> Tensor:
        return torch.stack([torch.linalg.matrix_norm(S) for S in self._factors]).prod()

    # inverse ------------------------------------------------------------------------------
    def inverse(
        self,
        damping: float = 0.0,
        use_heuristic_damping: bool = False,
        min_damping: float = 1e-8,
        use_exact_damping: bool = False,
        retry_double_precision: bool = True,
    ) -> PyTorchLinearOperator:
        ensure_all_square(*self._factors)
        fs = self._factors
        if use_heuristic_damping and use_exact_damping:
            raise ValueError("Either use heuristic damping or exact damping, not both.")
        if use_heuristic_damping and len(fs) > 2:
            raise ValueError(f"Heuristic damping only implemented for at most two factors. Got {len(fs)}")
        if use_exact_damping:
            # (S1 (x) S2 + d I)^-1 via the eigendecompositions of the (symmetric) factors
            evals, evecs = zip(*[linalg_native.eigh(S) for S in fs])
            lam = evals[0]
            for e in evals[1:]:
                lam = torch.kron(lam, e)
            return EighDecomposedLinearOperator(lam, KroneckerProductLinearOperator(*evecs)).inverse(damping=damping)
        if use_heuristic_damping and len(fs) == 1:
            dampings = (max(damping, min_damping),)
        elif use_heuristic_damping:
            S1, S2 = fs
            m1, m2 = S1.diag().mean(), S2.diag().mean()
            if m1 < 0 or m2 < 0:
                raise RuntimeError("Negative mean eigenvalue detected")
            pi = (m2 / m1).sqrt()
            root = sqrt(damping)
            dampings = (max(root / pi, min_damping), max(root * pi, min_damping))
        else:
            dampings = tuple(len(fs) * [damping])
        inv = [
            linalg_native.damped_cholesky_inverse(S, float(d), retry_double_precision)
            for S, d in zip(fs, dampings)
        ]
        return KroneckerProductLinearOperator(*inv)


class EighDecomposedLinearOperator(PyTorchLinearOperator):
    """``Q diag(lambda) Q^T`` with ``Q`` a dense matrix or a (Kronecker) operator."""

    SELF_ADJOINT: bool = True

    def __init__(self, eigenvalues: Tensor, eigenvectors: Tensor | PyTorchLinearOperator):
        if eigenvalues.ndim != 1:
            raise ValueError(f"Eigenvalues must be 1D, got shape {eigenvalues.shape}.")
        if len(eigenvectors.shape) != 2:
            raise ValueError(f"Eigenvectors must be 2D, got shape {eigenvectors.shape}.")
        if eigenvectors.shape[0] != eigenvectors.shape[1]:
            raise ValueError(f"Eigenvectors must be square, got shape {eigenvectors.shape}.")
        if eigenvalues.shape[0] != eigenvectors.shape[0]:
            raise ValueError(
                f"Incompatible shapes: eigenvalues {eigenvalues.shape}, eigenvectors {eigenvectors.shape}."
            )
        self._eigenvalues = eigenvalues
        self._eigenvectors = eigenvectors
        n = eigenvalues.shape[0]
        super().__init__([(n,)], [(n,)])

    @property
    def eigenvalues(self) -> Tensor:
        return self._eigenvalues

    @eigenvalues.setter
    def eigenvalues(self, value: Tensor) -> None:
        _expect_same_shape(self._eigenvalues, value)
        _expect_same_device(self._eigenvalues, value)
        _expect_same_dtype(self._eigenvalues, value)
        self._eigenvalues = value

    def _scale(self, x: Tensor) -> Tensor:
        lam = self._eigenvalues
        if is_kmajor(x):   # keep the K-major layout of the Kronecker basis products
            return (x.T * lam.unsqueeze(0)).T
        if is_native_tensor(x) and is_native_tensor(lam) and x.is_contiguous():
            return _hip.rowscale(x, lam.contiguous())
        return lam.unsqueeze(1) * x

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        (x,) = X
        Q = self._eigenvectors
        if isinstance(Q, Tensor):
            if is_native_tensor(Q) and is_native_tensor(x):
                QTx = _hip.gemm(Q.T, x.contiguous())
                return [_hip.gemm(Q, self._scale(QTx))]
            return [Q @ (self._eigenvalues.unsqueeze(1) * (Q.mH @ x))]
        K = x.shape[-1]
        if (type(Q) is KroneckerProductLinearOperator and len(Q) == 2 and is_native_tensor(x) and (K == 1 or is_kmajor(x))
                and all(is_native_tensor(f) and f.shape[0] == f.shape[1] for f in Q) and is_native_tensor(self._eigenvalues)):
            fc = _contiguous_pair(list(Q), False)   # (the eigensolver hands back row-eigenvector arrays: .T views)
            xk = x.T if K > 1 else x.reshape(1, -1)
            if fc is not None and xk.is_contiguous():   # Q1 (lam .* (Q1^T X Q2)) Q2^T in ONE foreign call
                y = _hip.eigh_apply(fc[0], fc[1], self._eigenvalues.contiguous(), xk, K, rows=fc[2])
                return [y.T if K > 1 else y.reshape(-1, 1)]
        (QTx,) = Q._adjoint_matmat([x])
        (res,) = Q._matmat([self._scale(QTx)])
        return [res]

    @property
    def device(self) -> torch.device:
        return infer_device([self._eigenvalues, self._eigenvectors])

    @property
    def dtype(self) -> torch.dtype:
        return infer_dtype([self._eigenvalues, self._eigenvectors])

    def trace(self) -> Tensor:
        return self._eigenvalues.sum()

    def det(self) -> Tensor:
        return self._eigenvalues.prod()

    def logdet(self) -> Tensor:
        return self._eigenvalues.log().sum()

    def frobenius_norm(self) -> Tensor:
        return self._eigenvalues.norm(p="fro")

    def inverse(self, damping: float = 0.0) -> "EighDecomposedLinearOperator":
        return EighDecomposedLinearOperator(1.0 / (self._eigenvalues + damping), self._eigenvectors)


class BlockDiagonalLinearOperator(PyTorchLinearOperator):
    """Block-diagonal operator; block ``i`` consumes its own slice of the tensor list."""

    def __init__(self, blocks: list[PyTorchLinearOperator]):
        if not blocks:
            raise ValueError("At least one block must be provided.")
        self._blocks = blocks
        in_shape = [tuple(s) for B in blocks for s in B._in_shape]
        out_shape = [tuple(s) for B in blocks for s in B._out_shape]
        super().__init__(in_shape, out_shape)
        self.SELF_ADJOINT = all(B.SELF_ADJOINT for B in blocks)

    def __iter__(self) -> Iterator[PyTorchLinearOperator]:
        return iter(self._blocks)

    def __len__(self) -> int:
        return len(self._blocks)

    def __getitem__(self, index: int) -> PyTorchLinearOperator:
        return self._blocks[index]

    def __setitem__(self, index: int, value: PyTorchLinearOperator) -> None:
        old = self._blocks[index]
        _expect_same_spaces(old, value)
        _expect_same_device(old, value)
        _expect_same_dtype(old, value)
        self._blocks[index] = value

    _POOL_STREAMS = 4

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        parts = split_list(X, [len(B._in_shape) for B in self._blocks])
        if len(self._blocks) >= 2 and all(is_native_tensor(x) for x in X):
            # single vectors of nets with repeated layer shapes: the equal blocks as ONE batched product pair fill the
            # chip better than one product per block (ResNet-18, tools/probe_kron_blocks.py: KFAC 1.32 vs 1.38 ms, EKFAC
            # 2.44 vs 2.78 ms); everything else that is a vector or K-major goes through ONE foreign call
            if len(self._blocks) >= 4 and X[0].shape[-1] == 1 and self.GROUP_FIRST:
                out = self._matmat_grouped(parts)
                if out is not None:
                    return out
            out = self._matmat_single_call(parts)
            if out is not None:
                return out
        if len(self._blocks) >= 4 and all(is_native_tensor(x) for x in X):
            return self._matmat_concurrent(parts)
        out: list[Tensor] = []
        for B, xs in zip(self._blocks, parts):
            out.extend(B._matmat(xs))
        return out

    SINGLE_CALL = True   # (tools/probe_kron_blocks.py flips these for A/B runs)
    GROUP_FIRST = True

    @property
    def assume_frozen(self) -> bool:
        """Kept for symmetry with the curvature operators; a block-diagonal operator holds NO copies of its blocks' factors
        any more (the batched equal-shape products reference them in place), so there is nothing to freeze or refresh."""
        return getattr(self, "_assume_frozen", False)

    @assume_frozen.setter
    def assume_frozen(self, value: bool) -> None:
        self._assume_frozen = bool(value)

    def _matmat_single_call(self, parts: list[list[Tensor]]) -> list[Tensor] | None:
        """All blocks ``S1 (x) S2`` / ``(Q1 (x) Q2) diag(lam) (Q1 (x) Q2)^T`` of a KFAC / EKFAC operator in ONE foreign
        call (``clo_kron_matmat_blocks``) when the operands are vectors or K-major blocks; None: take the other routes."""
        if not self.SINGLE_CALL or any(len(xs) != 1 for xs in parts):
            return None
        K = parts[0][0].shape[-1]
        pairs = [self._kron_pair(B) for B in self._blocks]
        if any(p is None for p in pairs):
            return None
        blocks, xs = [], []
        for (kp, lam), (x,) in zip(pairs, parts):
            if x.shape[-1] != K or not (K == 1 or is_kmajor(x)):
                return None
            xk = x.T if K > 1 else x.reshape(1, -1)
            fc = _contiguous_pair(list(kp), False)
            if fc is None or not xk.is_contiguous():
                return None
            if lam is not None:
                if fc[0].shape[0] != fc[0].shape[1] or fc[1].shape[0] != fc[1].shape[1]:
                    return None
                lam = lam.contiguous()
            blocks.append((fc[0], fc[1], lam, fc[2]))
            xs.append(xk)
        ys = _hip.kron_blocks(blocks, xs, K)
        return [y.T if K > 1 else y.reshape(-1, 1) for y in ys]

    @staticmethod
    def _kron_pair(B) -> tuple[KroneckerProductLinearOperator, Tensor | None] | None:
        """(two-factor native Kronecker operator, eigenvalues or None) of a groupable block."""
        lam = None
        if type(B) is EighDecomposedLinearOperator and is_native_tensor(B.eigenvalues):
            B, lam = B._eigenvectors, B.eigenvalues
        if type(B) is KroneckerProductLinearOperator and len(B) == 2 and all(is_native_tensor(f) for f in B):
            return B, lam
        return None

    def _kron_groups(self) -> list[tuple[list[int], list[Tensor], list[Tensor], list[Tensor] | None]]:
        """Blocks ``S1 (x) S2`` (KFAC) or ``(Q1 (x) Q2) diag(lambda) (Q1 (x) Q2)^T`` (EKFAC) with identical factor shapes
        and strides (repeated layer shapes): ``[(block indices, [S1_i], [S2_i], [lambda_i] | None), ...]``.  The factors
        are REFERENCED, not copied: the batched products take them where they lie (``_hip.gemm_members`` /
        ``clo_gemm_ptrs_f32``), so every product reads the live factor tensors like the reference's loop over blocks
        (``block_diagonal.py``) -- ``factor.data.mul_()``, an EMA update or a replaced factor need no refresh."""
        pairs = [self._kron_pair(B) for B in self._blocks]
        by_shape: dict = {}
        for i, p in enumerate(pairs):
            if p is not None:
                f1, f2 = p[0][0], p[0][1]
                if f1.data_ptr() % 16 or f2.data_ptr() % 16:
                    continue   # (the batched launch wants 16-byte aligned members; such a block takes the other routes)
                by_shape.setdefault((tuple(f1.shape), tuple(f2.shape), f1.stride(), f2.stride(), p[1] is None), []).append(i)
        groups = []
        for idx in by_shape.values():
            if len(idx) < 2:
                continue
            lam = None if pairs[idx[0]][1] is None else [pairs[i][1] for i in idx]
            groups.append((idx, [pairs[i][0][0] for i in idx], [pairs[i][0][1] for i in idx], lam))
        return groups

    def _matmat_grouped(self, parts: list[list[Tensor]]) -> list[Tensor] | None:
        """Single vectors: blocks of equal shape run as ONE batched product pair ``S1_i X_i S2_i^T``
        instead of 2 GEMMs each on a handful of CUs (no split-K slabs, no reduce launches: the batch
        fills the chip); the remaining blocks go through the stream pool."""
        groups = self._kron_groups()
        if not groups:
            return None
        results: dict[int, list[Tensor]] = {}
        for idx, S1, S2, lam in groups:
            if lam is None:   # Y_i = S1_i X_i S2_i^T
                a, b = S1[0].shape[1], S2[0].shape[1]
                Xb = torch.stack([parts[i][0].reshape(a, b) for i in idx])      # [n, a, b]
                Y = _hip.gemm_members(_hip.gemm_members(S1, Xb), [f.T for f in S2])    # [n, A, B]
            else:             # Y_i = Q1_i (lambda_i * (Q1_i^T X_i Q2_i)) Q2_i^T
                a, b = S1[0].shape[0], S2[0].shape[0]
                Xb = torch.stack([parts[i][0].reshape(a, b) for i in idx])
                Z = _hip.gemm_members(_hip.gemm_members([f.T for f in S1], Xb), S2)
                Z.mul_(torch.stack([l.reshape(S1[0].shape[1], S2[0].shape[1]) for l in lam]))
                Y = _hip.gemm_members(_hip.gemm_members(S1, Z), [f.T for f in S2])
            for k, i in enumerate(idx):
                results[i] = [Y[k].reshape(-1, 1)]
        rest = [i for i in range(len(self._blocks)) if i not in results]
        if len(rest) >= 4:
            sub = BlockDiagonalLinearOperator([self._blocks[i] for i in rest])
            ys = split_list(sub._matmat_concurrent([parts[i] for i in rest]), [len(self._blocks[i]._out_shape) for i in rest])
            for i, y in zip(rest, ys):
                results[i] = y
        else:
            for i in rest:
                results[i] = self._blocks[i]._matmat(parts[i])
        return [t for i in range(len(self._blocks)) for t in results[i]]

    def _matmat_concurrent(self, parts: list[list[Tensor]]) -> list[Tensor]:
        """The blocks are independent and individually too small to fill 256 CUs (a ResNet-18 KFAC
        product is 21 blocks of two GEMMs each): spread them over a few HIP streams and join."""
        dev = parts[0][0].device
        pool = [side_stream(dev, i) for i in range(self._POOL_STREAMS)]   # the package-wide worker streams
        main = torch.cuda.current_stream(dev)
        ready = main.record_event()
        for side in pool:
            side.wait_event(ready)
        out: list[Tensor] = []
        # largest blocks first on each stream would balance better; program order keeps it simple
        for i, (B, xs) in enumerate(zip(self._blocks, parts)):
            side = pool[i % len(pool)]
            with torch.cuda.stream(side):
                ys = B._matmat(xs)
            for t in (*xs, *ys):
                t.record_stream(side)
            out.extend(ys)
        for side in pool:
            main.wait_stream(side)
        for t in out:
            t.record_stream(main)
        return out

    def _adjoint(self) -> "BlockDiagonalLinearOperator":
        return BlockDiagonalLinearOperator([B.adjoint() for B in self._blocks])

    @property
    def device(self) -> torch.device:
        return infer_device(self._blocks)

    @property
    def dtype(self) -> torch.dtype:
        return infer_dtype(self._blocks)

    def trace(self) -> Tensor:
        ensure_all_square(*self._blocks)
        return torch.stack([B.trace() for B in self._blocks]).sum()

    def det(self) -> Tensor:
        ensure_all_square(*self._blocks)
        return torch.stack([B.det() for B in self._blocks]).prod()

    def logdet(self) -> Tensor:
        ensure_all_square(*self._blocks)
        return torch.stack([B.logdet() for B in self._blocks]).sum()

    def frobenius_norm(self) -> Tensor:
        return torch.stack([B.frobenius_norm() ** 2 for B in self._blocks]).sum().sqrt()
