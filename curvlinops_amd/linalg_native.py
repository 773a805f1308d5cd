"""Dense factor post-processing: damped Cholesky inverse and symmetric eigendecomposition.

``damped_cholesky_inverse`` reproduces ``KroneckerProductLinearOperator._damped_cholesky_inverse``
(reference ``curvlinops/kronecker.py:328-373``): out-of-place damping of the diagonal,
``cholesky`` + ``cholesky_inverse``, one retry in float64 (with a warning) when the
factorisation fails.  ``eigh`` replaces ``torch.linalg.eigh`` at ``kronecker.py:294`` and
``computers/_base.py:369-372``.

fp32 GPU inputs: the Cholesky inverse runs on the hand-written kernels (``csrc/linalg.hip`` for
the diagonal blocks, the MFMA GEMM of ``csrc/gemm.hip`` for every O(n^3) step; driver
``_hip.cholesky_inverse``).  The symmetric eigensolver is hand-written end to end: Householder reduction
``csrc/sytrd.hip`` (one persistent launch per 64-column panel), tridiagonal divide & conquer and block-reflector
back-transformation ``csrc/eigh.hip`` / ``eigh_native.py``.  ``torch.linalg.eigh`` remains for CPU / float64
operands (the reference's own path) and as the float64 retry of a result that fails its verification; the rocSOLVER
comparison routes of rounds 1-3 live in ``tools/`` (``tools/_rocsolver.py``, ``tools/probe_rocsolver_phases.py``).
"""

from __future__ import annotations

import os

from contextlib import contextmanager
from warnings import warn

import torch
from torch import Tensor

from curvlinops_amd import _hip
from curvlinops_amd.utils import is_native_tensor
from curvlinops_amd.utils import side_stream as _side_stream   # one pool of worker streams per device for the whole package


def _torch_damped_cholesky_inverse(A: Tensor, damping: float) -> Tensor:
    damped = torch.diagonal_scatter(A, A.diag() + damping)
    return torch.cholesky_inverse(torch.linalg.cholesky(damped))


class _InverseBatch:
    """Factor inverses are dependency-bound chains of small kernels (a few hundred launches for a
    4608 x 4608 factor) that leave most of the chip idle.  The jobs are collected; factors of EQUAL
    size then share one chain (``clo_cholesky_inverse_batched_f32``: one workgroup per factor in the
    leaves, batched GEMMs), the chains of different sizes are driven largest-first by a few worker
    threads that each own a HIP stream (one foreign call per chain, GIL released), and all pivot
    statuses are inspected with ONE device read at the end.

    ``distributed=True`` (factors replicated on every rank, as after the KFAC factor all-reduce):
    the jobs are additionally sharded over the ranks, largest-first, and the inverses exchanged
    with one packed broadcast per owner (``dist.partition_by_cost``)."""

    MIN_THREADED = 3     # fewer jobs than this: run them inline on the caller's stream
    MIN_TOTAL_N = 2048   # ... as are batches of tiny factors (LeNet-5: threads cost more than they hide)

    def __init__(self, num_streams: int, distributed: bool = False):
        self._num = num_streams
        self.distributed = distributed
        self._jobs: list = []  # [A, damping, retry, out, status (device int32 | None), error]

    def submit(self, A: Tensor, damping: float, retry: bool) -> Tensor:
        n = A.shape[0]
        native = is_native_tensor(A)
        out = torch.empty(n, n, device=A.device, dtype=A.dtype)
        status = torch.zeros(1, device=A.device, dtype=torch.int32) if native else None
        self._jobs.append([A, damping, retry, out, status, None])
        return out

    @staticmethod
    def _run_job(job: list) -> None:
        A, damping, _, out, status, _ = job
        if status is not None:
            _hip.cholesky_inverse_into(A, damping, out, status)
            return
        try:  # non-native tensors (CPU, float64): the torch path, failure kept for the retry logic
            out.copy_(_torch_damped_cholesky_inverse(A, damping))
        except RuntimeError as error:
            job[5] = error

    MAX_GROUP_BYTES = 8 << 30  # workspace bound of one batched call

    def _units(self, native: list) -> list[list]:
        """Factors of equal size share ONE chain of launches (batched leaves / GEMMs); split only to
        bound the workspace."""
        by_n: dict[int, list] = {}
        for job in native:
            by_n.setdefault(job[0].shape[0], []).append(job)
        units = []
        for n, group in by_n.items():
            per = 5 * 4 * ((n + 3) // 4 * 4) ** 2 + (4 << 20)
            chunk = max(1, self.MAX_GROUP_BYTES // per)
            units.extend(group[i : i + chunk] for i in range(0, len(group), chunk))
        units.sort(key=lambda u: -len(u) * u[0][0].shape[0] ** 3)
        return units

    @staticmethod
    def _run_unit(unit: list) -> None:
        if len(unit) == 1:
            _InverseBatch._run_job(unit[0])
            return
        status = torch.zeros(len(unit), device=unit[0][0].device, dtype=torch.int32)
        _hip.cholesky_inverse_batched_into([j[0] for j in unit], [j[1] for j in unit], [j[3] for j in unit], status)
        for i, job in enumerate(unit):
            job[4] = status[i : i + 1]

    def _run(self, jobs: list) -> None:
        native = [j for j in jobs if j[4] is not None]
        for job in jobs:
            if job[4] is None:
                self._run_job(job)
        units = self._units(native)
        if (len(units) < self.MIN_THREADED or self._num < 2
                or sum(len(u) * u[0][0].shape[0] for u in units) < self.MIN_TOTAL_N):
            for unit in units:
                self._run_unit(unit)
            return
        import queue
        import threading

        device = native[0][0].device
        main = torch.cuda.current_stream(device)
        ready = main.record_event()
        todo: queue.SimpleQueue = queue.SimpleQueue()
        for unit in units:
            todo.put(unit)
        done: list = []
        errors: list = []

        def worker(index: int) -> None:
            try:
                with torch.cuda.device(device):
                    side = _side_stream(device, index)
                    side.wait_event(ready)
                    with torch.cuda.stream(side):
                        while True:
                            try:
                                unit = todo.get_nowait()
                            except queue.Empty:
                                break
                            for job in unit:
                                for t in (job[0], job[3], job[4]):
                                    t.record_stream(side)
                            self._run_unit(unit)
                            for job in unit:
                                job[4].record_stream(side)
                    done.append(side.record_event())
            except BaseException as e:  # noqa: BLE001 -- re-raised in the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(min(self._num, len(units)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for ev in done:
            main.wait_event(ev)
        if errors:
            raise errors[0]

    def finish(self) -> None:
        jobs = self._jobs
        if not jobs:
            return
        from curvlinops_amd import dist as cdist

        sharded = self.distributed and cdist.is_distributed()
        if sharded:
            import torch.distributed as tdist

            owner = cdist.partition_by_cost([float(j[0].shape[0]) ** 3 for j in jobs], tdist.get_world_size())
            rank = tdist.get_rank()
            mine = [j for j, o in zip(jobs, owner) if o == rank]
        else:
            mine = jobs
        self._run(mine)
        # failures of this rank's jobs: retry in float64 or remember the error
        native = [j for j in mine if j[4] is not None]
        if native:
            for job, pivot in zip(native, torch.cat([j[4] for j in native]).cpu().tolist()):
                if pivot:
                    job[5] = _hip.not_pd_error(pivot, job[0].shape[0])
        fatal = None
        for job in mine:
            A, damping, retry, out, _, error = job
            if error is None:
                continue
            if not retry or A.dtype == torch.float64:
                fatal = fatal or error
                continue
            _warn_retry(A, error)
            out.copy_(_torch_damped_cholesky_inverse(A.to(torch.float64), damping))
        if sharded:
            flag = torch.tensor([1.0 if fatal is not None else 0.0], device=jobs[0][0].device)
            tdist.all_reduce(flag, op=tdist.ReduceOp.MAX)
            if fatal is None and float(flag) > 0:
                fatal = RuntimeError("cholesky: a factor owned by another rank is not positive-definite.")
        if fatal is not None:
            raise fatal
        if sharded:
            world = tdist.get_world_size()
            for r in range(world):
                theirs = [j for j, o in zip(jobs, owner) if o == r]
                if not theirs:
                    continue
                ref = theirs[0][3]
                if r == rank:
                    flat = torch.cat([j[3].reshape(-1).to(ref.dtype) for j in theirs])
                else:
                    flat = torch.empty(sum(j[3].numel() for j in theirs), device=ref.device, dtype=ref.dtype)
                tdist.broadcast(flat, src=r)
                if r != rank:
                    off = 0
                    for j in theirs:
                        n = j[3].numel()
                        j[3].copy_(flat[off : off + n].view_as(j[3]))
                        off += n


_ACTIVE_BATCH: _InverseBatch | None = None


# default number of worker streams of concurrent_inverses: ONE since round 4 -- every big factor's call is already a pipeline
# over a helper stream of its own; in the bench process 1 / 2 / 4 workers take 11.8 - 12.1 / 14.1 - 15.0 / 14.2 ms for ResNet-18's
# 42 factors (tools/run_inv_workers.sh; round 3, with four hardware queues, had two ahead)
# Round 6: THREE again, with the equal-size groups on the plain batched chain (csrc/linalg.hip: the per-group pipeline and its
# host-timed helper-stream probe are gone): 10.3 - 11.5 ms for ResNet-18's 42 factors at 4 and 16 hardware queues, first call
# included (one worker 13.7 - 14.1, two 10.8 - 11.6; profiles/r06_cholesky_streams.txt).
INVERSE_WORKERS = int(os.environ.get("CLO_INV_WORKERS", "3"))   # (the environment variable: A/B runs only)


@contextmanager
def concurrent_inverses(num_streams: int | None = None, distributed: bool = False):
    """Inside the block, fp32 GPU calls of :func:`damped_cholesky_inverse` are collected and return
    their (still empty) output tensors immediately; on exit the factors are inverted concurrently
    (worker threads with their own streams -- ONE by default, `INVERSE_WORKERS`: the big factors' calls are pipelines over a helper stream
    of their own, and more than a handful of busy HIP streams share hardware queues: ResNet-18's 42 factors 9.2-9.5 ms
    with two workers, 10-12 with four, 19-30 ms with the equal-size groups split into concurrent single calls --),
    failed factorisations are redone in float64 into the
    SAME output tensor (or raise, as in the synchronous form).  ``distributed=True``: the factors
    are replicated on all ranks of the default process group and the work is sharded by factor."""
    global _ACTIVE_BATCH
    if _ACTIVE_BATCH is not None:  # nested: the outermost block owns the batch
        yield
        return

    batch = _ACTIVE_BATCH = _InverseBatch(INVERSE_WORKERS if num_streams is None else num_streams, distributed)
    try:
        yield
    except BaseException:
        _ACTIVE_BATCH = None
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        raise
    else:
        _ACTIVE_BATCH = None
        batch.finish()


def _warn_retry(A: Tensor, error: Exception) -> None:
    warn(
        f"Failed to compute Cholesky decomposition in {A.dtype} precision with error {error}. "
        "Retrying in double precision...",
        stacklevel=3,
    )


def damped_cholesky_inverse(A: Tensor, damping: float, retry_double_precision: bool = True) -> Tensor:
    """``(A + damping I)^-1`` for symmetric positive definite ``A`` (never modifies ``A``)."""
    native = is_native_tensor(A) and _hip.has("clo_potrf_diag_f32")
    if _ACTIVE_BATCH is not None and (native or _ACTIVE_BATCH.distributed):
        return _ACTIVE_BATCH.submit(A, damping, retry_double_precision)
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


_SYTRD_MAX_N = 8184   # the reduction keeps two vectors of the matrix order in LDS


def _unit_scale(A: Tensor) -> tuple[Tensor, Tensor]:
    """``(A / s, s)`` with ``s = max |A|`` per matrix (1 for a zero matrix), no host synchronisation.
    rocSOLVER's tridiagonal divide & conquer (``sstedc``, also inside ``torch.linalg.eigh``) applies an ABSOLUTE
    tolerance: fp32 matrices of norm 1e-5 come back with eigenvalue errors of 6 % of the largest one, norm 1e-6
    with 30 % (``tools/diag_eigh_scale.py``; the Householder reduction before it is scale-invariant).  Gradient
    covariances of mean-reduced losses live at exactly those scales, so every GPU eigensolver call here runs
    on the normalised matrix and scales the eigenvalues back (what LAPACK's ``syev`` does for norms outside
    its safe range)."""
    s = A.abs().amax(dim=(-2, -1), keepdim=True)
    s = torch.where(s > 0, s, torch.ones_like(s))
    return A / s, s


# healthy float32 eigenvectors: |Q^T Q - I| <= 2e-5 up to order 8000 (full-rank "Wishart" factors 1 ... 4e-6).  Covariances with
# large clusters of (near-)equal eigenvalues -- dead ReLU units, repeated rows: tests/golden/eigh_regression.npz -- can come
# back with 5e-5 ... 1e-4 from the divide & conquer; they are redone in float64 (the bound was 1e-4 while the absolute
# residual test still rejected those cases for another reason)
_ORTH_TOL = 4e-5


def _orth_defect(Q: Tensor) -> Tensor:
    """``max |Q^T Q - I|`` per matrix (device tensor)."""
    n = Q.shape[-1]
    G = Q.mT @ Q
    G.diagonal(dim1=-2, dim2=-1).sub_(1.0)
    return G.abs().amax(dim=(-2, -1)) if n > 0 else G.new_zeros(G.shape[:-2])


# Acceptance of a float32 eigendecomposition: |A Q - Q diag(lam)| <= _RES_C * eps32 * ||A||_F, the backward-error form
# (a backward-stable solver leaves residual columns of norm p(n) eps ||A||_2; measured on the hand-written route,
# tools/diag_eigh_verify.py: 2 ... 40 eps ||A||_F for low-rank and full-rank covariances of order 577 ... 4609).  The
# bound scales with the matrix: the absolute 1e-3 on the max-normalised matrix used before rejected CORRECT results of
# well-conditioned (full-rank) factors, whose norm is ~0.3 n max|A|, and sent them to the float64 vendor solver.
_RES_C = 400.0
_EPS32 = 1.1920929e-07
FLOAT64_RETRIES = 0   # eigendecompositions that failed the float32 acceptance test and were redone in float64 (tests / bench read it)


def _note_float64_retry(count: int = 1) -> None:
    global FLOAT64_RETRIES
    FLOAT64_RETRIES += count


def _residual_defect(An: Tensor, lam: Tensor, Q: Tensor) -> Tensor:
    """``max |A Q - Q diag(lam)|`` per matrix (device tensor)."""
    return (An @ Q - Q * lam.unsqueeze(-2)).abs().amax(dim=(-2, -1))


def _residual_tol(An: Tensor) -> Tensor:
    """Per-matrix acceptance bound ``_RES_C eps32 ||A||_F`` (device tensor, no host synchronisation)."""
    return (_RES_C * _EPS32) * torch.linalg.matrix_norm(An.float(), ord="fro", dim=(-2, -1))


def _torch_eigh_scaled(A: Tensor) -> tuple[Tensor, Tensor]:
    """``torch.linalg.eigh`` on the normalised matrices; on the GPU in float32 the eigenvectors are VERIFIED:
    rocSOLVER's ``ssyevd`` returns non-orthogonal eigenvectors (|Q^T Q - I| = 0.07 ... 0.27, residual fine) for
    some covariances of ReLU features with dead units and repeated rows (found by tools/fuzz_kfac.py, factor
    kept in tests/golden/eigh_regression.npz).  A matrix that fails the check is decomposed again with the
    hand-written reduction (``eigh_sytrd``), and in float64 if that is not applicable."""
    if not A.is_cuda:
        res = torch.linalg.eigh(A)
        return res.eigenvalues, res.eigenvectors
    An, s = _unit_scale(A)
    res = torch.linalg.eigh(An)
    lam, Q = res.eigenvalues, res.eigenvectors
    if A.dtype == torch.float32 and A.shape[-1] > 1:
        ok = (_orth_defect(Q) <= _ORTH_TOL) & (_residual_defect(An, lam, Q) <= _residual_tol(An))   # (NaN fails both)
        bad = (~ok).reshape(-1).nonzero().flatten().tolist()
        if bad:
            batch_shape = An.shape[:-2]
            An2, lam2, Q2 = An.reshape(-1, *An.shape[-2:]), lam.reshape(-1, lam.shape[-1]).clone(), Q.reshape(-1, *Q.shape[-2:]).clone()
            for b in bad:
                lam2[b], Q2[b] = _eigh_unit_checked(An2[b])
            lam, Q = lam2.reshape(*batch_shape, -1), Q2.reshape(*batch_shape, *Q.shape[-2:])
    return lam * s.squeeze(-1), Q


def _eigh_unit_checked(An: Tensor) -> tuple[Tensor, Tensor]:
    """Second opinion for ONE normalised float32 GPU matrix: own reduction, then float64."""
    n = An.shape[0]
    if 3 <= n <= _SYTRD_MAX_N:
        try:
            lam, Q = _eigh_sytrd_unit(An)
            if bool((_orth_defect(Q) <= _ORTH_TOL) & (_residual_defect(An, lam, Q) <= _residual_tol(An))):
                return lam, Q
        except (RuntimeError, OSError):   # solver did not converge / library or LDS attribute unavailable
            pass
    _note_float64_retry()
    res = torch.linalg.eigh(An.double())
    return res.eigenvalues.float(), res.eigenvectors.float()


def _eigh_sytrd_unit(An: Tensor, max_blocks: int = 0) -> tuple[Tensor, Tensor]:
    """``clo_eigh_f32`` on a normalised matrix (no checks)."""
    n = An.shape[0]
    ld = (n + 3) // 4 * 4
    work = torch.zeros(1, n, ld, device=An.device, dtype=torch.float32)   # zero padding columns
    work[0, :, :n].copy_(An)
    lam, Z = _hip.eigh_batched_(work, n, max_blocks)
    return lam[0], Z[0, :, :n].T


def eigh_sytrd(A: Tensor, max_blocks: int = 0) -> tuple[Tensor, Tensor]:
    """Symmetric eigendecomposition on the hand-written kernels: ``clo_sytrd_f32`` (persistent panel launches) ->
    Cuppen divide & conquer -> block-reflector back-transformation, verified (orthogonality, residual) with a float64
    retry.  fp32 GPU matrices of order 3..8184; conventions of ``torch.linalg.eigh`` (ascending eigenvalues,
    eigenvectors in columns)."""
    n = A.shape[0]
    if not (A.is_cuda and A.dtype == torch.float32 and A.dim() == 2 and A.shape[1] == n and 3 <= n <= _SYTRD_MAX_N):
        raise ValueError(f"eigh_sytrd: need a square fp32 GPU matrix of order 3..{_SYTRD_MAX_N}, got {tuple(A.shape)} {A.dtype}")
    return _eigh_native_group([A], max_blocks)[0]


def _eigh_native_group(As: list[Tensor], max_blocks: int = 0) -> list[tuple[Tensor, Tensor]]:
    """Hand-written route for several fp32 GPU matrices of ONE order n >= 3 (repeated layer shapes): ONE foreign call
    (``clo_eigh_batched_f32``: one reduction per matrix, a divide & conquer whose tree levels carry all matrices, one
    back-transformation per matrix -- no Python between the stages), then a batched verification (own GEMMs) with the
    float64 retry of the reference's Cholesky path as a model."""
    n, B = As[0].shape[0], len(As)
    dev = As[0].device
    An, scale = _unit_scale(torch.stack(As))
    ld = (n + 3) // 4 * 4
    work = torch.zeros(B, n, ld, device=dev, dtype=torch.float32)
    work[:, :, :n] = An
    lam, Z = _hip.eigh_batched_(work, n, max_blocks)
    Zc = Z[:, :, :n]
    Q = Zc.mT
    # verification on the engine: |Q^T Q - I| and |A Q - Q diag(lam)| per matrix, one host read for the group.  Joint weight +
    # bias factors have odd orders (577 ... 4609): contracted over the PADDED row length ld (zeros behind column n on both
    # sides) the products take the engine's float4 loaders instead of the scalar ones (n = 4609: 2 x 196 GFLOP)
    if ld != n:
        Z[:, :, n:].zero_()
        An_p = torch.zeros(B, n, ld, device=dev, dtype=torch.float32)
        An_p[:, :, :n] = An
    else:
        An_p = An
    G = _hip.gemm(Z, Z.mT)                                    # rows of Z are the eigenvectors
    G.diagonal(dim1=-2, dim2=-1).sub_(1.0)
    R = _hip.gemm(An_p, Z.mT) - Q * lam.unsqueeze(-2)
    ok = ((G.abs().amax(dim=(-2, -1)) <= _ORTH_TOL) & (R.abs().amax(dim=(-2, -1)) <= _residual_tol(An))).tolist()
    out = []
    for b in range(B):
        if ok[b]:
            out.append((lam[b] * scale[b].reshape(()), Q[b]))
        else:
            _note_float64_retry()
            res = torch.linalg.eigh(An[b].double())
            out.append((res.eigenvalues.float() * scale[b].reshape(()), res.eigenvectors.float()))
    return out


def _nonzero_rows(A: Tensor) -> Tensor | None:
    """Indices of the rows (= columns) of symmetric ``A`` that are not entirely zero, or None if all are.
    Covariances of ReLU features have exactly-zero rows for dead units (ResNet-18, 512 CIFAR-sized inputs: 4180
    of the 4608 patch features of a layer4 convolution): the eigenproblem splits exactly into the nonzero
    principal submatrix and unit vectors with eigenvalue 0, which is both much smaller and the input class on
    which rocSOLVER's ``ssyevd`` loses orthogonality (|Q^T Q - I| = 0.59 on that factor)."""
    nz = (A != 0).any(dim=1)
    m = int(nz.sum())
    return None if m == A.shape[0] else nz.nonzero().flatten()


def _embed_deflated(n: int, idx: Tensor, lam_s: Tensor, Q_s: Tensor) -> tuple[Tensor, Tensor]:
    """Eigenpairs of the full matrix from those of its nonzero principal submatrix (rows ``idx``)."""
    m = idx.numel()
    dev, dt = lam_s.device, lam_s.dtype
    lam = torch.cat([lam_s, lam_s.new_zeros(n - m)])
    Q = torch.zeros(n, n, device=dev, dtype=dt)
    if m:
        Q[idx.unsqueeze(1), torch.arange(m, device=dev).unsqueeze(0)] = Q_s
    dead = torch.ones(n, dtype=torch.bool, device=dev)
    dead[idx] = False
    dead_idx = dead.nonzero().flatten()
    Q[dead_idx, m + torch.arange(n - m, device=dev)] = 1.0
    order = torch.sort(lam, stable=True).indices
    return lam[order], Q[:, order]


def _eigh_2x2(A: Tensor) -> tuple[Tensor, Tensor]:
    """Closed form for a symmetric 2 x 2 matrix (the G factors of two-class heads): the Jacobi rotation that
    annihilates the off-diagonal entry, evaluated in float64 with device-side elementwise ops (no host
    synchronisation, no iteration that could fail to converge); ascending eigenvalues, orthonormal columns."""
    a, b, c = A[0, 0].double(), 0.5 * (A[0, 1].double() + A[1, 0].double()), A[1, 1].double()
    theta = 0.5 * torch.atan2(2.0 * b, a - c)
    cs, sn = torch.cos(theta), torch.sin(theta)
    lam1 = a * cs * cs + 2.0 * b * cs * sn + c * sn * sn
    lam2 = a * sn * sn - 2.0 * b * cs * sn + c * cs * cs
    Q = torch.stack([torch.stack([cs, -sn]), torch.stack([sn, cs])])          # columns: (cs, sn), (-sn, cs)
    swap = lam1 > lam2
    lam = torch.where(swap, torch.stack([lam2, lam1]), torch.stack([lam1, lam2]))
    Q = torch.where(swap, Q.flip(1), Q)
    return lam.to(A.dtype), Q.to(A.dtype)


def _eigh_full(A: Tensor, max_blocks: int = 0) -> tuple[Tensor, Tensor]:
    """One matrix without zero rows: fp32 GPU matrices on the hand-written solver, everything else (CPU, float64,
    orders beyond the reduction's 8184) through ``torch.linalg.eigh`` on the normalised matrix."""
    if A.is_cuda and A.dtype == torch.float32 and A.dim() == 2:
        n = A.shape[0]
        if 3 <= n <= _SYTRD_MAX_N:
            return eigh_sytrd(A, max_blocks)
        if n == 2:   # one Jacobi rotation in float64: exact, nothing to verify
            return _eigh_2x2(A)
    return _torch_eigh_scaled(A)


def eigh(A: Tensor) -> tuple[Tensor, Tensor]:
    """Eigenvalues (ascending) and orthonormal eigenvectors (columns) of symmetric ``A``."""
    if A.is_cuda and A.dim() == 2 and A.shape[0] > 1:
        idx = _nonzero_rows(A)
        if idx is not None:
            if idx.numel() == 0:
                return A.new_zeros(A.shape[0]), torch.eye(A.shape[0], device=A.device, dtype=A.dtype)
            sub = A.index_select(0, idx).index_select(1, idx)
            return _embed_deflated(A.shape[0], idx, *_eigh_full(sub))
    return _eigh_full(A)


EIGH_WORKERS = 12   # default number of worker streams of eigh_many (module attribute: tools/run_workers.sh sweeps it)


def eigh_many(mats: list[Tensor], num_streams: int | None = None) -> list[tuple[Tensor, Tensor]]:
    """:func:`eigh` of several independent symmetric matrices.  A single decomposition is a chain of dependent
    steps (one grid-wide hand-off per matrix column in the reduction), so

    * factors of EQUAL size share the divide & conquer levels and the verification (networks repeat layer shapes), and
    * the units are spread, largest first, over a few worker threads that each own a HIP stream; the persistent
      panel launches of the reductions that run side by side share the chip's 256 CUs in proportion to their matrix
      sizes (``max_blocks`` of ``clo_sytrd_f32``; the library's admission control keeps any combination safe)."""
    if num_streams is None:
        # twelve workers (six until round 4): the reduction is latency-bound per column, so more factors in flight -- each on a
        # share of the CUs -- shorten the set (tools/probe_eigh_streams.py, ResNet-18's 42 factors, 6 / 8 / 12 / 16 / 20
        # workers: 92 / 83 / 76 / 71 / 73 ms with one hardware queue per stream, 119 / 98 / 96 / 100 / 105 ms with the
        # runtime's default of four)
        num_streams = EIGH_WORKERS
    out: list = [None] * len(mats)
    for i, A in enumerate(mats):
        if not A.is_cuda:
            out[i] = eigh(A)
    # exactly-zero rows (dead ReLU features) split off: the solvers see the nonzero principal submatrices
    full_mats, deflated = mats, {}
    mats = list(mats)
    cand = [i for i, A in enumerate(full_mats) if A.is_cuda and A.dim() == 2 and A.shape[0] > 1 and out[i] is None]
    masks = {i: (full_mats[i] != 0).any(dim=1) for i in cand}
    # one device -> host read for all factors (a read per factor stalled the launch stream ~40 times per refresh)
    counts = dict(zip(cand, torch.stack([masks[i].sum() for i in cand]).tolist())) if cand else {}
    for i in cand:
        A = full_mats[i]
        if counts[i] == A.shape[0]:
            continue
        if counts[i] == 0:
            out[i] = (A.new_zeros(A.shape[0]), torch.eye(A.shape[0], device=A.device, dtype=A.dtype))
        else:
            idx = deflated[i] = masks[i].nonzero().flatten()
            mats[i] = A.index_select(0, idx).index_select(1, idx)
    gpu = [i for i, A in enumerate(mats) if A.is_cuda and out[i] is None]
    if not gpu:
        return out
    out = _eigh_many_gpu(mats, gpu, out, num_streams)
    for i, idx in deflated.items():
        out[i] = _embed_deflated(full_mats[i].shape[0], idx, *out[i])
    return out


def _eigh_many_gpu(mats: list[Tensor], gpu: list[int], out: list, num_streams: int) -> list:
    groups: dict = {}
    for i in gpu:
        groups.setdefault((mats[i].shape[0], mats[i].dtype), []).append(i)
    units: list[list[int]] = []

    # Wall-time estimate of one decomposition (ms), fitted to tools/probe_eigh_sizes.py on MI355X (dense full-rank input:
    # 2.8 / 5.9 / 10.8 / 25.8 / 71.7 / 248 ms at n = 64 / 256 / 576 / 1152 / 2304 / 4608; rank-deficient factors are faster
    # by a common factor).  n^3 (the flop count) put all four 2304-factors of a ResNet-18 into one unit "worth" half a
    # 4608-factor, although they take 1.2 x as long.
    def est(n: int) -> float:
        return 2.5 + 0.008 * n + 9e-6 * float(n) ** 2

    total = sum(est(mats[i].shape[0]) for i in gpu)
    largest = max(est(mats[i].shape[0]) for i in gpu)
    target = max(largest, total / max(num_streams, 1))  # no unit longer than the best possible makespan
    for (n, dtype), idx in groups.items():
        per = 8 * max(n, 1) ** 2 * mats[idx[0]].element_size()  # stacked input + vectors + solver workspace
        fit = max(1, int(target / max(est(n), 1e-9)))           # matrices of this size one worker can take
        parts = -(-len(idx) // fit)
        chunk = max(1, min((8 << 30) // per, -(-len(idx) // parts)))
        units.extend(idx[k : k + chunk] for k in range(0, len(idx), chunk))
    units.sort(key=lambda u: -len(u) * est(mats[u[0]].shape[0]))
    # workgroups per panel launch: the units that start together (the `num_streams` largest) split the 256 CUs by
    # matrix area, later (smaller) units take the share their size would have had among those
    lead = sum(mats[u[0]].shape[0] ** 2 for u in units[: max(num_streams, 1)])
    blocks = {id(u): int(min(256, max(16, round(256.0 * mats[u[0]].shape[0] ** 2 / max(lead, 1))))) for u in units}
    if len(units) == 1:
        blocks[id(units[0])] = 0   # a lone reduction: the library's default (every CU it can use)

    def run(unit: list[int]) -> None:
        mb = blocks[id(unit)]
        if len(unit) == 1:
            for i in unit:
                out[i] = _eigh_full(mats[i], mb)
            return
        A0 = mats[unit[0]]
        if A0.dtype == torch.float32 and 3 <= A0.shape[0] <= _SYTRD_MAX_N:
            for i, res in zip(unit, _eigh_native_group([mats[i] for i in unit], mb)):
                out[i] = res
            return
        lam, vec = _torch_eigh_scaled(torch.stack([mats[i] for i in unit]))
        for k, i in enumerate(unit):
            out[i] = (lam[k], vec[k])

    if len(units) < 2 or num_streams < 2:
        for unit in units:
            run(unit)
        return out
    import queue
    import threading

    jobs: queue.SimpleQueue = queue.SimpleQueue()
    for unit in units:
        jobs.put(unit)
    device = mats[gpu[0]].device
    main = torch.cuda.current_stream(device)
    ready = main.record_event()
    done: list = []
    errors: list = []

    def worker(index: int) -> None:
        try:
            with torch.cuda.device(device):
                side = _side_stream(device, index)
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    while True:
                        try:
                            unit = jobs.get_nowait()
                        except queue.Empty:
                            break
                        for i in unit:
                            mats[i].record_stream(side)
                        run(unit)
                done.append(side.record_event())
        except BaseException as e:  # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(min(num_streams, len(units)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    for ev in done:
        main.wait_event(ev)
    for res in out:
        for t in res:
            if t.is_cuda:
                t.record_stream(main)
    return out
