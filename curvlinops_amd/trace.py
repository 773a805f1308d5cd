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
from curvlinops_amd.utils import assert_divisible_by, assert_is_square, assert_matvecs_subseed_dim, is_native_tensor

_QR_MAX_ELEMS = 2**30  # rocSOLVER / BLAS index with 32-bit ints: keep every library call below 2^31


def frobenius_inner(X: Tensor, Y: Tensor) -> Tensor:
    """``sum_ij X_ij Y_ij`` (the reference's ``einsum("ij,ij", X, Y)``).  fp32 GPU blocks use the
    64-bit-indexed streaming kernel ``clo_dot_f32`` (one pass over both operands, no temporary);
    ``[D, K]`` blocks with ``D K >= 2^31`` (C5) are beyond what the BLAS-backed einsum accepts."""
    if is_native_tensor(X) and is_native_tensor(Y) and X.is_contiguous() and Y.is_contiguous():
        return _hip.dot(X, Y)
    return torch.einsum("ij,ij", X, Y)


_GRAM_MIN_ELEMS = 2**22  # above this, fp32 GPU blocks use the GEMM-rich Gram route


def _gram64(X: Tensor) -> Tensor:
    """``X^T X`` in float64 (exact products, float64 accumulation) for a tall float32 GPU block."""
    if _hip.tall_gram_supported(X):
        return _hip.tall_gram(X)
    gram = torch.empty(X.shape[1], X.shape[1], device=X.device, dtype=torch.float32)   # wider than 64 columns
    _hip.syrk_accum(gram, X, alpha=1.0, beta=0.0)
    return gram.double()


def _times_small(X: Tensor, T: Tensor) -> Tensor:
    """``X T`` for a tall block and a small square ``T`` (float64 in, rounded once)."""
    T32 = T.float().contiguous()
    if _hip.tall_apply_supported(X, T32):
        return _hip.tall_apply(X, T32)
    return _hip.gemm(X, T32)


def project_out(Q: Tensor, G: Tensor) -> Tensor:
    """``G - Q (Q^T G)`` (reference ``meyer2020hutch.py:97-99``): on float32 GPU blocks two streaming passes --
    ``clo_tall_gram_f64`` for the coefficients (float64 accumulation over the 1e7 ... 1e8 rows), ``clo_tall_apply_f32`` for
    the update -- instead of two library GEMMs with a [D, N] temporary between them."""
    if _hip.tall_gram_supported(Q, G):
        if Q.shape[1] % 4:   # (the update kernel reads Q in 16-byte pieces: zero columns change nothing)
            Qp = Q.new_zeros(Q.shape[0], -(-Q.shape[1] // 4) * 4)
            Qp[:, : Q.shape[1]] = Q
            Q = Qp
        C = _hip.tall_gram(Q, G).neg_().float().contiguous()
        if _hip.tall_apply_supported(Q, C, G):
            return _hip.tall_apply(Q, C, G, beta=1.0)
    return G - Q @ (Q.T @ G)


# Relative eigenvalue of the float64 Gram matrix below which a direction of the block is numerically DEPENDENT: sigma below
# 3e-6 of the largest singular value is inside the rounding noise of a float32 block that came out of an operator product
# (~1e-6 sigma_max after the O(D) float32 accumulations) -- exact rank deficiency (repeated / zero columns, a rank-8 operator
# sketched with 32 probes) sits at 1e-14.  Such directions are not range information: `orthonormal_basis` replaces them by
# random directions (n columns, as Householder QR returns), Hutch++ leaves them out -- for a symmetric operator whose range
# the kept columns contain they contribute exactly zero to both terms of the estimator, and with a decaying spectrum what
# they could carry is < 1e-11 of the trace -- so the third operator product runs on r <= n columns.
_RANK_TOL = 1e-11


def _gram_orthonormal_basis(X: Tensor, complete: bool = True) -> Tensor:
    """Orthonormal basis with ALL ``n`` columns (as the reference's Householder ``Q``, ``meyer2020hutch.py:89-93``) of a very
    tall float32 GPU block from two Gram passes: ``Q = X V diag(lambda)^-1/2`` with ``X^T X = V diag(lambda) V^T``, then
    once more on ``Q`` to push the loss of orthogonality to eps.  The Gram matrices are accumulated in FLOAT64 from exact
    products (``clo_tall_gram_f64``), so directions are resolved down to the rounding noise of the float32 data
    (sigma / sigma_max ~ 1e-7) -- a float32 Gram matrix loses everything below sqrt(eps) ~ 3e-3, and round 5 dropped
    those directions from the basis, which made the estimator differ from the reference's on decaying spectra.
    Directions that are exactly dependent (relative eigenvalue < 1e-13) are replaced by random directions orthogonalised
    against the rest: Householder QR also returns n orthonormal columns for a rank-deficient block, and Hutch++ is exact
    for any orthonormal basis that contains range(X).  Everything O(m) is a streaming kernel (Householder QR of an
    [85M, 32] block takes seconds in the vendor solver, this takes tens of milliseconds)."""
    Q = X if X.is_contiguous() else X.contiguous()
    n = Q.shape[1]
    missing = 0
    for it in range(4):
        gram = _gram64(Q)
        # (normalised: rocSOLVER's tridiagonal solver applies an absolute tolerance, linalg_native._unit_scale)
        gscale = gram.abs().amax().clamp_min(torch.finfo(torch.float32).tiny)
        lam, V = torch.linalg.eigh(gram / gscale)
        lam = lam * gscale
        keep = lam > lam.max() * _RANK_TOL
        kept = int(keep.sum())
        if kept == 0:
            Q, missing = Q[:, :0], n
            break
        if kept < Q.shape[1]:
            missing = n - kept
        Q = _times_small(Q, V[:, keep] / lam[keep].sqrt())
        # the rounding of X T in float32 leaves |Q^T Q - I| ~ eps32 * cond(input): a pass whose INPUT was already
        # well conditioned has produced an orthonormal block (two passes for any sketch that is not nearly singular,
        # a third for spectra that reach the float32 noise floor)
        if it >= 1 and float(lam[keep].min() / lam.max()) > 0.25:
            break
    if not complete and Q.shape[1] == 0:
        return torch.zeros(X.shape[0], 1, device=X.device, dtype=X.dtype)   # (a zero block: one zero column, no range)
    if missing and complete:
        # complete the basis: random directions, twice projected off range(Q), orthonormalised among themselves
        R = torch.randn(X.shape[0], missing, device=X.device, dtype=X.dtype)
        if Q.shape[1]:
            for _ in range(2):
                R = project_out(Q, R)
        R = _gram_orthonormal_basis(R.contiguous())
        Q = torch.cat([Q, R], dim=1) if Q.shape[1] else R
    return Q


def _gram_qr(X: Tensor) -> tuple[Tensor, Tensor] | None:
    """``(Q, T^-T)`` with ``X = Q T``, ``Q`` orthonormal, for a tall full-rank fp32 GPU block -- the factorisation
    XTrace / XDiag need (``trace/epperly2024xtrace.py:52-60`` call ``torch.linalg.qr``, which is rocSOLVER on this
    platform).  Their leave-one-out vectors ``s_i`` are the directions of ``T^-T e_i``: the complement of
    ``range(X[:, != i])`` inside ``range(Q)`` whatever the shape of ``T``, so ``T`` need not be triangular.  Two Gram
    passes (``clo_tall_gram_f64``: float64 accumulation; ``clo_tall_apply_f32``), the two small eigenproblems in float64;
    ``T = (L2^1/2 V2^T)(L1^1/2 V1^T)`` is inverted in closed form.  None if ``X`` is numerically rank-deficient (the
    caller then takes the float64 route)."""
    Q = X if X.is_contiguous() else X.contiguous()
    n = Q.shape[1]
    Tinv = torch.eye(n, device=X.device, dtype=torch.float64)
    for it in range(2):
        gram = _gram64(Q)
        gscale = gram.abs().amax().clamp_min(torch.finfo(torch.float32).tiny)
        lam, V = torch.linalg.eigh(gram / gscale)
        lam = lam * gscale
        if not bool((lam > lam.max() * (1e-10 if it == 0 else 1e-12)).all()):
            return None
        step = V / lam.sqrt()                      # X_it = X_{it+1} (L^1/2 V^T)  =>  X_{it+1} = X_it (V L^-1/2)
        Q = _times_small(Q, step)
        Tinv = Tinv @ step
    return Q, Tinv.T.to(X.dtype)


def orthonormal_basis(X: Tensor, complete: bool = True) -> Tensor:
    """``Q`` of the reduced QR factorisation of a tall ``[m, n]`` matrix (``meyer2020hutch.py:93``).
    Above ``2^30`` elements the factorisation is done as TSQR -- Householder QR of row chunks, QR of
    the stacked triangular factors, one small GEMM per chunk -- which is as stable as the direct
    call, works for rank-deficient inputs, and keeps every library call within 32-bit indexing."""
    m, n = X.shape
    if is_native_tensor(X) and m >= 2 * n:   # fp32 on the GPU: always the Gram route on the own GEMM engine
        return _gram_orthonormal_basis(X, complete)
    if m * n <= _QR_MAX_ELEMS or m <= 2 * n:
        return torch.linalg.qr(X)[0]
    rows = max(2 * n, _QR_MAX_ELEMS // n)
    bounds = list(range(0, m, rows)) + [m]
    if len(bounds) > 2 and bounds[-1] - bounds[-2] < n:  # last chunk too short for a reduced QR
        bounds.pop(-2)
    Qs, Rs = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        q, r = torch.linalg.qr(X[lo:hi])
        Qs.append(q)
        Rs.append(r)
    Q2 = torch.linalg.qr(torch.cat(Rs, dim=0))[0]
    out = torch.empty_like(X)
    for i, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        torch.matmul(Qs[i], Q2[i * n : (i + 1) * n], out=out[lo:hi])
        Qs[i] = None
    return out


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
    return frobenius_inner(G, A @ G) / num_matvecs


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
    Q = orthonormal_basis(A @ S, complete=False)   # (numerically dependent directions left out: see `_RANK_TOL`)
    tr_range = frobenius_inner(Q, A @ Q)
    G = random_matrix(dim, N, distribution, dev, dt) if probes is None else probes[1]
    AG = A @ project_out(Q, G)
    AG = project_out(Q, AG)
    return tr_range + frobenius_inner(G, AG) / N


def hutchinson_diag(A: Tensor | PyTorchLinearOperator, num_matvecs: int, distribution: str = "rademacher",
                    probes: Tensor | None = None) -> Tensor:
    """Hutchinson estimator of the diagonal, ``mean_k g_k * (A g_k)`` (reference
    ``diagonal/hutchinson.py``); probes packed as for the trace estimators."""
    dim = assert_is_square(A)
    assert_matvecs_subseed_dim(A, num_matvecs)
    G = random_matrix(dim, num_matvecs, distribution, A.device, A.dtype) if probes is None else probes
    return (G * (A @ G)).sum(dim=1) / num_matvecs


def hutchinson_squared_fro(A: Tensor | PyTorchLinearOperator, num_matvecs: int, distribution: str = "rademacher",
                           probes: Tensor | None = None) -> Tensor:
    """Hutchinson estimator of the squared Frobenius norm, ``mean_k ||A g_k||^2`` (reference
    ``norm/hutchinson.py``); a wide matrix is applied through its transpose."""
    if len(A.shape) != 2:
        raise ValueError(f"A must be a matrix. Got shape {A.shape}.")
    dim = min(A.shape)
    if num_matvecs >= dim:
        raise ValueError(f"num_matvecs ({num_matvecs}) must be less than the minimum dimension of A.")
    if A.shape[1] > A.shape[0]:
        A = A.T if isinstance(A, Tensor) else A.adjoint()
    G = random_matrix(dim, num_matvecs, distribution, A.device, A.dtype) if probes is None else probes
    AG = A @ G
    return frobenius_inner(AG, AG) / num_matvecs


def _leave_one_out_vectors(R: Tensor) -> Tensor:
    """Columns ``s_i`` with ``Q_i Q_i^T = Q (I - s_i s_i^T) Q^T`` for the bases ``Q_i`` one would get
    from the QR factorisation without the i-th test vector (Epperly et al. 2024)."""
    RT_inv = torch.linalg.inv(R.T)
    return RT_inv / (RT_inv**2).sum(0) ** 0.5


def _q_and_leave_one_out(A_W: Tensor) -> tuple[Tensor, Tensor]:
    """``(Q, S)`` for XTrace / XDiag: fp32 GPU blocks through :func:`_gram_qr` (own kernels), everything else --
    and numerically rank-deficient blocks -- through the reference's QR (in float64 for the fallback)."""
    if is_native_tensor(A_W) and A_W.shape[0] >= 2 * A_W.shape[1]:
        qt = _gram_qr(A_W)
        if qt is not None:
            Q, RT_inv = qt
            return Q, RT_inv / (RT_inv**2).sum(0) ** 0.5
        Q, R = torch.linalg.qr(A_W.double())
        return Q.to(A_W.dtype), _leave_one_out_vectors(R).to(A_W.dtype)
    Q, R = torch.linalg.qr(A_W)
    return Q, _leave_one_out_vectors(R)


def xtrace(A: Tensor | PyTorchLinearOperator, num_matvecs: int, distribution: str = "rademacher",
           probes: Tensor | None = None) -> Tensor:
    """XTrace (Epperly, Tropp & Webber 2024; reference ``trace/epperly2024xtrace.py``): exchangeable
    leave-one-out combination of a low-rank trace and Hutchinson on the complement.  The loop over
    test vectors of the reference is evaluated for all of them at once (three small GEMMs)."""
    dim = assert_is_square(A)
    assert_matvecs_subseed_dim(A, num_matvecs)
    assert_divisible_by(num_matvecs, 2, "num_matvecs")
    N = num_matvecs // 2
    W = random_matrix(dim, N, distribution, A.device, A.dtype) if probes is None else probes
    A_W = A @ W
    Q, S = _q_and_leave_one_out(A_W)
    A_Q = A @ Q
    QT_A_Q = Q.T @ A_Q
    traces = QT_A_Q.trace() - torch.einsum("ij,ik,kj->j", S, QT_A_Q, S)
    # (I - Q_i Q_i^T) A (I - Q_i Q_i^T) w_i for all i; deflation = v - <s_i, v> s_i column by column
    def deflate(V: Tensor) -> Tensor:
        return V - S * (S * V).sum(0)

    A_P_W = A_W - A_Q @ deflate(Q.T @ W)
    PT_A_P_W = A_P_W - Q @ deflate(Q.T @ A_P_W)
    return (traces + (W * PT_A_P_W).sum(0)).mean()


def xdiag(A: Tensor | PyTorchLinearOperator, num_matvecs: int, probes: Tensor | None = None) -> Tensor:
    """XDiag (reference ``diagonal/epperly2024xtrace.py``; Rademacher test vectors): the diagonal
    counterpart of :func:`xtrace`.  Needs products with ``A^T`` (``Q^T A``)."""
    dim = assert_is_square(A)
    assert_matvecs_subseed_dim(A, num_matvecs)
    assert_divisible_by(num_matvecs, 2, "num_matvecs")
    N = num_matvecs // 2
    W = random_matrix(dim, N, "rademacher", A.device, A.dtype) if probes is None else probes
    A_W = A @ W
    Q, S = _q_and_leave_one_out(A_W)
    QT_A = Q.T @ A
    diagonal = (Q * QT_A.T).sum(1) - ((Q @ S) * (QT_A.T @ S)).sum(1) / N

    def deflate(V: Tensor) -> Tensor:
        return V - S * (S * V).sum(0)

    A_comp_W = A_W - Q @ deflate(QT_A @ W)
    return diagonal + (W * A_comp_W / W**2).sum(1) / N

