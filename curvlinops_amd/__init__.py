"""curvlinops_amd -- MI355X (gfx950) native backend for the curvature-matvec hot path of
f-dangel/curvlinops, behind the reference's linear-operator API.

Public names mirror ``curvlinops/__init__.py`` for the path in scope:
curvature operators, KFAC / EKFAC with their structured building blocks, and the probe-packing
trace estimators.  The compute kernels live in ``csrc/*.hip`` (C ABI in
``include/curvlinops_amd.h``); importing this package does not need a GPU, running a GPU fp32
operator does need ``lib/libclo_hip.so`` (build with ``python -m curvlinops_amd.csrc.build``).
"""

from curvlinops_amd.canonical import FromCanonicalLinearOperator, ToCanonicalLinearOperator
from curvlinops_amd.curvature import (
    CurvatureLinearOperator,
    EFLinearOperator,
    GGNLinearOperator,
    HessianLinearOperator,
)
from curvlinops_amd.diag import DiagonalLinearOperator
from curvlinops_amd.enums import FisherType, KFACType
from curvlinops_amd.ggn_diagonal import GGNDiagonalLinearOperator
from curvlinops_amd.inverse import (
    CGInverseLinearOperator,
    LSMRInverseLinearOperator,
    NeumannInverseLinearOperator,
)
from curvlinops_amd.jacobian import JacobianLinearOperator, TransposedJacobianLinearOperator
from curvlinops_amd.kfac import EKFACLinearOperator, KFACLinearOperator
from curvlinops_amd.kfoc import KFOCLinearOperator
from curvlinops_amd.kronecker import (
    BlockDiagonalLinearOperator,
    EighDecomposedLinearOperator,
    KroneckerProductLinearOperator,
)
from curvlinops_amd.linop import PyTorchLinearOperator
from curvlinops_amd.trace import (
    hutchinson_diag,
    hutchinson_squared_fro,
    hutchinson_trace,
    hutchpp_trace,
    xdiag,
    xtrace,
)

__all__ = [
    "PyTorchLinearOperator",
    "CurvatureLinearOperator",
    "HessianLinearOperator",
    "GGNLinearOperator",
    "EFLinearOperator",
    "JacobianLinearOperator",
    "TransposedJacobianLinearOperator",
    "DiagonalLinearOperator",
    "GGNDiagonalLinearOperator",
    "CGInverseLinearOperator",
    "LSMRInverseLinearOperator",
    "NeumannInverseLinearOperator",
    "KFACLinearOperator",
    "EKFACLinearOperator",
    "KFOCLinearOperator",
    "KroneckerProductLinearOperator",
    "EighDecomposedLinearOperator",
    "BlockDiagonalLinearOperator",
    "ToCanonicalLinearOperator",
    "FromCanonicalLinearOperator",
    "FisherType",
    "KFACType",
    "hutchinson_trace",
    "hutchpp_trace",
    "hutchinson_diag",
    "hutchinson_squared_fro",
    "xtrace",
    "xdiag",
]
