"""Operator algebra of the drop-in boundary: ``PyTorchLinearOperator`` and its sum / scale /
chain composites.

Contract reproduced from the reference (``curvlinops/_torch_base.py:33-814``):
an operator maps a tensor-product space with shapes ``_in_shape`` to one with ``_out_shape``;
``A @ X`` accepts a flat ``[N]`` / ``[N, K]`` tensor or a tensor list ``[*N_i]`` / ``[*N_i, K]``
and returns the same format; ``X @ A`` uses leading ``K``; ``A @ B`` builds a flattened chain;
subclasses implement ``_matmat(list[Tensor]) -> list[Tensor]`` with the column axis TRAILING.
Composites call ``_matmat`` directly, so a natively implemented block only has to provide it.
"""

from __future__ import annotations

from collections.abc import Callable, Iterator
from dataclasses import dataclass

import numpy
import torch
from scipy.sparse.linalg import LinearOperator
from torch import Size, Tensor

from curvlinops_amd.utils import allclose_report


@dataclass
class _Format:
    """How the user passed the operand (restored on the way out)."""

    as_list: bool
    is_vector: bool
    num_cols: int


def _expect_same_spaces(old: "PyTorchLinearOperator", new: "PyTorchLinearOperator") -> None:
    if old._in_shape != new._in_shape or old._out_shape != new._out_shape:
        raise ValueError(
            f"Shape mismatch: expected in_shape={old._in_shape}, out_shape={old._out_shape}, "
            f"got in_shape={new._in_shape}, out_shape={new._out_shape}."
        )


def _expect_same_device(old, new) -> None:
    if old.device != new.device:
        raise ValueError(f"Device mismatch: expected {old.device}, got {new.device}.")


def _expect_same_dtype(old, new) -> None:
    if old.dtype != new.dtype:
        raise ValueError(f"Dtype mismatch: expected {old.dtype}, got {new.dtype}.")


def _expect_same_shape(old: Tensor, new: Tensor) -> None:
    if old.shape != new.shape:
        raise ValueError(f"Shape mismatch: expected {old.shape}, got {new.shape}.")


def _adjacent_rows(Y: list[Tensor], sizes: list[int], K: int) -> Tensor | None:
    """The ``[sum(sizes), K]`` matrix the blocks of ``Y`` already form when they are consecutive row
    ranges of one contiguous buffer (a producer that wrote its result in place); None otherwise."""
    base = Y[0]._base
    if base is None or base.dim() != 2 or base.shape != (sum(sizes), K) or not base.is_contiguous():
        return None
    # only buffers a producer allocated for this very result (tagged by ``fresh_result_buffer``):
    # a pass-through operator may hand back views of the CALLER's matrix, which must never be
    # returned as "our" result (composites scale / accumulate results in place)
    if not getattr(base, "_clo_fresh_result", False):
        return None
    off = base.storage_offset()
    for y, n in zip(Y, sizes):
        if y._base is not base or not y.is_contiguous() or y.storage_offset() != off:
            return None
        off += n * K
    return base


def fresh_result_buffer(rows: int, K: int, device, dtype) -> Tensor:
    """A ``[rows, K]`` result buffer whose row-range views ``_from_list`` may return without a copy."""
    buf = torch.empty(rows, K, device=device, dtype=dtype)
    buf._clo_fresh_result = True
    return buf


def _shares_storage(a: Tensor, b) -> bool:
    return isinstance(b, Tensor) and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()


class PyTorchLinearOperator:
    """Linear operator on tensor-product spaces of PyTorch tensors.

    Attributes:
        SELF_ADJOINT: if True, ``adjoint()`` returns ``self`` and ``_adjoint`` is not needed.
    """

    SELF_ADJOINT: bool = False

    def __init__(self, in_shape: list[tuple[int, ...]], out_shape: list[tuple[int, ...]]):
        if not in_shape or not out_shape:
            raise ValueError(f"In- {in_shape} and output shapes {out_shape} must be non-empty.")
        self._in_shape = [Size(s) for s in in_shape]
        self._out_shape = [Size(s) for s in out_shape]
        self._in_shape_flat = [s.numel() for s in self._in_shape]
        self._out_shape_flat = [s.numel() for s in self._out_shape]
        self.shape = (sum(self._out_shape_flat), sum(self._in_shape_flat))

    # ------------------------------------------------------------------ to implement
    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        """Multiply onto ``[*N_1, K], [*N_2, K], ...``; return ``[*M_1, K], ...``."""
        raise NotImplementedError

    def _adjoint(self) -> "PyTorchLinearOperator":
        raise NotImplementedError

    @property
    def device(self) -> torch.device:
        raise NotImplementedError

    @property
    def dtype(self) -> torch.dtype:
        raise NotImplementedError

    def adjoint(self) -> "PyTorchLinearOperator":
        return self if self.SELF_ADJOINT else self._adjoint()

    # ------------------------------------------------------------------ format handling
    @staticmethod
    def _to_list(X, shapes: list[Size], leading: bool) -> tuple[list[Tensor], _Format]:
        """Validate ``X`` and bring it into matrix tensor-list format (explicit column axis,
        trailing unless ``leading``)."""
        sizes = [s.numel() for s in shapes]
        total = sum(sizes)
        if isinstance(X, Tensor):
            fixed = -1 if leading else 0
            if X.ndim not in (1, 2) or X.shape[fixed] != total:
                want = f"({total},) or " + (f"(K, {total})" if leading else f"({total}, K)")
                raise ValueError(
                    f"Input tensor must have shape {want}, with K arbitrary. Got {X.shape}."
                )
            is_vec = X.ndim == 1
            K = 1 if is_vec else X.shape[0 if leading else 1]
            pieces = X.split(sizes, dim=fixed)
            if leading:
                out = [p.reshape(K, *s) for p, s in zip(pieces, shapes)]
            else:
                out = [p.reshape(*s, K) for p, s in zip(pieces, shapes)]
            return out, _Format(False, is_vec, K)
        if isinstance(X, list) and all(isinstance(x, Tensor) for x in X):
            if len(X) != len(shapes):
                raise ValueError(f"Input list must have {len(shapes)} tensors. Got {len(X)}.")
            if all(x.shape == s for x, s in zip(X, shapes)):
                axis = 0 if leading else -1
                return [x.unsqueeze(axis) for x in X], _Format(True, True, 1)
            col_axis = 0 if leading else -1

            def body(x):
                return x.shape[1:] if leading else x.shape[:-1]

            ok = all(x.ndim == len(s) + 1 and body(x) == s for x, s in zip(X, shapes))
            cols = {x.shape[col_axis] for x in X} if ok else set()
            if not ok or len(cols) != 1:
                where = "leading" if leading else "trailing"
                raise ValueError(
                    f"Input list must contain tensors with shapes {shapes} and optional {where} "
                    f"dimension for the matrix columns. Got {[x.shape for x in X]}."
                )
            return list(X), _Format(True, False, cols.pop())
        raise ValueError(f"Input must be tensor or list of tensors. Got {type(X)}.")

    @staticmethod
    def _from_list(Y: list[Tensor], fmt: _Format, shapes: list[Size], leading: bool):
        """Check a result in matrix tensor-list format and restore the user's format."""
        if len(Y) != len(shapes):
            raise ValueError(f"Output tensor list must have {len(shapes)} tensors. Got {len(Y)}.")
        K = fmt.num_cols
        for y, s in zip(Y, shapes):
            want = (K, *s) if leading else (*s, K)
            if tuple(y.shape) != want:
                where = "leading" if leading else "trailing"
                raise ValueError(
                    f"Output tensors must have shapes {shapes} and additional {where} dimension "
                    f"of {K}. Got {[t.shape for t in Y]}."
                )
        axis = 0 if leading else -1
        if fmt.as_list:
            return [y.squeeze(axis) for y in Y] if fmt.is_vector else Y
        sizes = [s.numel() for s in shapes]
        if leading:
            flat = torch.cat([y.reshape(K, n) for y, n in zip(Y, sizes)], dim=1)
        else:
            flat = _adjacent_rows(Y, sizes, K)
            if flat is None:
                flat = torch.cat([y.reshape(n, K) for y, n in zip(Y, sizes)], dim=0)
        return flat.squeeze(axis) if fmt.is_vector else flat

    # ------------------------------------------------------------------ products
    def __matmul__(self, X):
        if isinstance(X, PyTorchLinearOperator):
            left = tuple(self) if isinstance(self, _ChainPyTorchLinearOperator) else (self,)
            right = tuple(X) if isinstance(X, _ChainPyTorchLinearOperator) else (X,)
            return _ChainPyTorchLinearOperator(*left, *right)
        Xl, fmt = self._to_list(X, self._in_shape, leading=False)
        return self._from_list(self._matmat(Xl), fmt, self._out_shape, leading=False)

    def __rmatmul__(self, X):
        # X @ A = (A^H X^H)^H ; X carries the column axis in front
        Xl, fmt = self._to_list(X, self._out_shape, leading=True)
        XH = [x.conj().movedim(0, -1) for x in Xl]
        YH = self.adjoint()._matmat(XH)
        Y = [y.conj().movedim(-1, 0) for y in YH]
        return self._from_list(Y, fmt, self._in_shape, leading=True)

    # ------------------------------------------------------------------ composition
    def __add__(self, other: "PyTorchLinearOperator") -> "_SumPyTorchLinearOperator":
        return _SumPyTorchLinearOperator(self, other)

    def __sub__(self, other: "PyTorchLinearOperator") -> "_SumPyTorchLinearOperator":
        return self + (-1.0 * other)

    def __mul__(self, scalar: int | float) -> "_ScalePyTorchLinearOperator":
        return _ScalePyTorchLinearOperator(self, scalar)

    def __rmul__(self, scalar: int | float) -> "_ScalePyTorchLinearOperator":
        return self * scalar

    def __truediv__(self, scalar: int | float) -> "_ScalePyTorchLinearOperator":
        return self * (1.0 / scalar)

    # ------------------------------------------------------------------ SciPy export
    def to_scipy(self, dtype: numpy.dtype | None = None) -> LinearOperator:
        """SciPy ``LinearOperator`` whose products run through this operator
        (numpy -> device tensor -> ``@`` -> host numpy, reference ``_torch_base.py:491-592``)."""
        dev, dt = self.device, self.dtype
        AH = self.adjoint()
        fwd = self._numpy_bridge(self.__matmul__, dev, dt)
        bwd = AH._numpy_bridge(AH.__matmul__, dev, dt)
        return LinearOperator(
            self.shape, matvec=fwd, rmatvec=bwd, matmat=fwd, rmatmat=bwd,
            dtype=numpy.dtype(dtype) if dtype is None else dtype,
        )

    @staticmethod
    def _numpy_bridge(f: Callable[[Tensor], Tensor], device, dtype):
        def g(X: numpy.ndarray) -> numpy.ndarray:
            Y = f(torch.as_tensor(X, dtype=dtype, device=device))
            if Y.dtype == torch.bfloat16:  # numpy has no bf16
                Y = Y.float()
            out = Y.detach().cpu().numpy().astype(X.dtype)
            if Y.is_cuda:   # the copy synchronised with the device: a launch that timed out is reported HERE, not later
                from curvlinops_amd import _hip

                _hip.raise_if_async_fault(Y.device.index)
            return out

        return g

    def _check_deterministic_matvec(self, rtol: float = 1e-5, atol: float = 1e-8) -> None:
        v = torch.rand(self.shape[1], device=self.device, dtype=self.dtype)
        if not allclose_report(self @ v, self @ v, rtol=rtol, atol=atol):
            raise RuntimeError("Check for deterministic matvec failed.")


class _SumPyTorchLinearOperator(PyTorchLinearOperator):
    """``A + B``."""

    def __init__(self, A: PyTorchLinearOperator, B: PyTorchLinearOperator):
        _expect_same_spaces(A, B)
        _expect_same_device(A, B)
        _expect_same_dtype(A, B)
        super().__init__(A._in_shape, A._out_shape)
        self._A, self._B = A, B
        self.SELF_ADJOINT = A.SELF_ADJOINT and B.SELF_ADJOINT

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        return [a + b for a, b in zip(self._A._matmat(X), self._B._matmat(X))]

    def __matmul__(self, X):
        # flat operands go through the summands' own `@` (and thus their fast paths), not through
        # the tensor-list detour
        if isinstance(X, Tensor) and X.dim() in (1, 2):
            a, b = self._A @ X, self._B @ X
            # in place only on memory that is provably not the caller's (inputs are never mutated,
            # reference `_torch_base.py:937`)
            return a + b if _shares_storage(a, X) else a.add_(b)
        return super().__matmul__(X)

    def _adjoint(self) -> "_SumPyTorchLinearOperator":
        return _SumPyTorchLinearOperator(self._A.adjoint(), self._B.adjoint())

    @property
    def device(self):
        return self._A.device

    @property
    def dtype(self):
        return self._A.dtype


class _ScalePyTorchLinearOperator(PyTorchLinearOperator):
    """``c * A``."""

    def __init__(self, A: PyTorchLinearOperator, scalar: float | int):
        super().__init__(A._in_shape, A._out_shape)
        self._A, self._scalar = A, scalar
        self.SELF_ADJOINT = A.SELF_ADJOINT

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        return [self._scalar * y for y in self._A._matmat(X)]

    def __matmul__(self, X):
        if isinstance(X, Tensor) and X.dim() in (1, 2):
            y = self._A @ X
            return self._scalar * y if _shares_storage(y, X) else y.mul_(self._scalar)
        return super().__matmul__(X)

    def _adjoint(self) -> "_ScalePyTorchLinearOperator":
        return _ScalePyTorchLinearOperator(self._A.adjoint(), self._scalar)

    @property
    def device(self):
        return self._A.device

    @property
    def dtype(self):
        return self._A.dtype


class _ChainPyTorchLinearOperator(PyTorchLinearOperator):
    """``A @ B @ C @ ...`` applied right to left."""

    def __init__(self, *operators: PyTorchLinearOperator):
        if len(operators) < 2:
            raise ValueError(f"Need at least 2 operators, got {len(operators)}.")
        for left, right in zip(operators[:-1], operators[1:]):
            if left._in_shape != right._out_shape:
                raise ValueError(
                    f"Shape mismatch: input shape {left._in_shape} does not match"
                    f" output shape {right._out_shape}."
                )
            _expect_same_device(left, right)
            _expect_same_dtype(left, right)
        self._operators = list(operators)
        super().__init__(operators[-1]._in_shape, operators[0]._out_shape)

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        for op in reversed(self._operators):
            X = op._matmat(X)
        return X

    def _adjoint(self) -> "_ChainPyTorchLinearOperator":
        return _ChainPyTorchLinearOperator(*(op.adjoint() for op in reversed(self._operators)))

    @property
    def device(self):
        return self._operators[0].device

    @property
    def dtype(self):
        return self._operators[0].dtype

    def __iter__(self) -> Iterator[PyTorchLinearOperator]:
        return iter(self._operators)

    def __len__(self) -> int:
        return len(self._operators)

    def __getitem__(self, index: int) -> PyTorchLinearOperator:
        return self._operators[index]

    def __setitem__(self, index: int, value: PyTorchLinearOperator) -> None:
        old = self._operators[index]
        _expect_same_spaces(old, value)
        _expect_same_device(old, value)
        _expect_same_dtype(old, value)
        self._operators[index] = value
