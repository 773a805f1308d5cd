"""Randomised trace estimators that drive an operator with PACKED probe matrices ``[D, K]``.

Algorithms as in the reference (``curvlinops/trace/hutchinson.py:13-75``,
``trace/meyer2020hutch.py:15-102``, ``sampling.py:6-56``).  On fp32 GPU operators the probes are
generated directly in the packed K-trailing layout by one counter-based Philox kernel
(``clo_pack_probes_f32``) instead of K separate RNG launches plus ``column_stack``; pass
``probes=...`` to inject fixed probe matrices (used for parity tests against the reference).
"""

from __future__ import annotations

import torch
from torch import Tensor

from curvlinops_amd import _hip
from curvlinops_amd.linop import PyTorchLinearOperator
from curvlinops_amd.utils import assert_divisible_by, assert_is_square, assert_matvecs_subseed_dim


def rademacher(dim: int, device, dtype) -> Tensor:
    return torch.empty(dim, device=device, dtype=dtype).bernoulli_(0.5).mul_(2).sub_(1)


def normal(dim: int, device, dtype) -> Tensor:
    return torch.randn(dim, device=device, dtype=dtype)


def random_vector(dim: int, distribution: str, device, dtype) -> Tensor:
    if distribution == "rademacher":
        return rademacher(dim, device, dtype)
    if distribution == "normal":
        return normal(dim, device, dtype)
    raise ValueError(f"Unknown distribution {distribution!r}.")


def random_matrix(dim: int, num: int, distribution: str, device, dtype) -> Tensor:
    """Packed ``[dim, num]`` probe matrix."""
    if distribution not in ("rademacher", "normal"):
        raise ValueError(f"Unknown distribution {distribution!r}.")
    dev = torch.device(device)
    if dev.type == "cuda" and dtype == torch.float32:
        seed = int(torch.randint(0, 2**62, (1,)).item())  # ties the stream to torch's global RNG
        return _hip.pack_probes(dim, num, seed, distribution, dev)
    return torch.column_stack([random_vector(dim, distribution, device, dtype) for _ in range(num)])


def hutchinson_trace(A: Tensor | PyTorchLinearOperator, num_matvecs: int, distribution: str = "rademacher",
                     probes: Tensor | None = None) -> Tensor:
    """Girard-Hutchinson estimator ``mean_k g_k^T A g_k``."""
    dim = assert_is_square(A)
    assert_matvecs_subseed_dim(A, num_matvecs)
    G = random_matrix(dim, num_matvecs, distribution, A.device, A.dtype) if probes is None else probes
    return torch.einsum("ij,ij", G, A @ G) / num_matvecs


def hutchpp_trace(A: Tensor | PyTorchLinearOperator, num_matvecs: int, distribution: str = "rademacher",
                  probes: tuple[Tensor, Tensor] | None = None) -> Tensor:
    """Hutch++ (Meyer et al. 2020): exact trace on the range of ``A S`` plus Hutchinson on the
    deflated remainder; three operator products with ``num_matvecs / 3`` columns each."""
    dim = assert_is_square(A)
    assert_matvecs_subseed_dim(A, num_matvecs)
    assert_divisible_by(num_matvecs, 3, "num_matvecs")
    N = num_matvecs // 3
    dev, dt = A.device, A.dtype
    S = random_matrix(dim, N, distribution, dev, dt) if probes is None else probes[0]
    Q, _ = torch.linalg.qr(A @ S)
    tr_range = torch.einsum("ji,ji", Q, A @ Q)
    G = random_matrix(dim, N, distribution, dev, dt) if probes is None else probes[1]
    AG = A @ (G - Q @ (Q.T @ G))
    AG = AG - Q @ (Q.T @ AG)
    return tr_range + torch.einsum("ij,ij", G, AG) / N
