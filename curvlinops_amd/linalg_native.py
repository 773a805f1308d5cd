"""Dense factor post-processing: damped Cholesky inverse and symmetric eigendecomposition.

``damped_cholesky_inverse`` reproduces ``KroneckerProductLinearOperator._damped_cholesky_inverse``
(reference ``curvlinops/kronecker.py:328-373``): out-of-place damping of the diagonal,
``cholesky`` + ``cholesky_inverse``, one retry in float64 (with a warning) when the
factorisation fails.  ``eigh`` replaces ``torch.linalg.eigh`` at ``kronecker.py:294`` and
``computers/_base.py:369-372``.

fp32 GPU inputs: the Cholesky inverse runs on the hand-written kernels (``csrc/linalg.hip`` for
the diagonal blocks, the MFMA GEMM of ``csrc/gemm.hip`` for every O(n^3) step; driver
``_hip.cholesky_inverse``).  The symmetric eigensolver is NOT native yet: ``eigh`` calls
``torch.linalg.eigh`` on the tensor's device (rocSOLVER on the GPU) -- see DESIGN.md, open items.
"""

from __future__ import annotations

from warnings import warn

import torch
from torch import Tensor

from curvlinops_amd import _hip
from curvlinops_amd.utils import is_native_tensor


def _torch_damped_cholesky_inverse(A: Tensor, damping: float) -> Tensor:
    damped = torch.diagonal_scatter(A, A.diag() + damping)
    return torch.cholesky_inverse(torch.linalg.cholesky(damped))


def damped_cholesky_inverse(A: Tensor, damping: float, retry_double_precision: bool = True) -> Tensor:
    """``(A + damping I)^-1`` for symmetric positive definite ``A`` (never modifies ``A``)."""
    native = is_native_tensor(A) and _hip.has("clo_potrf_diag_f32")
    try:
        if native:
            return _hip.cholesky_inverse(A, damping)
        return _torch_damped_cholesky_inverse(A, damping)
    except RuntimeError as error:
        if not retry_double_precision or A.dtype == torch.float64:
            raise error
        warn(
            f"Failed to compute Cholesky decomposition in {A.dtype} precision with error {error}. "
            "Retrying in double precision...",
            stacklevel=2,
        )
        return _torch_damped_cholesky_inverse(A.to(torch.float64), damping).to(A.dtype)


def eigh(A: Tensor) -> tuple[Tensor, Tensor]:
    """Eigenvalues (ascending) and orthonormal eigenvectors (columns) of symmetric ``A``."""
    if is_native_tensor(A) and _hip.has("clo_eigh_f32"):
        return _hip.eigh(A)
    res = torch.linalg.eigh(A)
    return res.eigenvalues, res.eigenvectors
