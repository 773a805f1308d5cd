"""Data-parallel sharding of the hot path: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on ROCm); ``gloo`` on CPU for tests.

Everything on the path is a normalised SUM over data (``_torch_base.py:939-942``,
``computers/kfac_hooks.py:350-353, 390-393``, ``computers/ekfac_hooks.py:456-458`` of the
reference), and the normalisation uses the GLOBAL number of data points.  So every rank builds
its operator / computer on its own shard of the mini-batches with ``num_data=<global N>`` and the
per-rank results are combined by ONE all-reduce(sum) of a packed buffer:

* matvec:  ``AllReducedLinearOperator(op)`` -- packed ``[D, K]`` result, ``4 D K`` bytes;
* KFAC:    ``allreduce_flat_(buffer)``  -- all ``A_l, G_l`` are views into one pre-allocated flat
  buffer that the SYRKs accumulate into and the collective reduces in place;
* EKFAC:   the same for the corrected eigenvalues.

After the factor all-reduce every rank holds IDENTICAL factors, so the O(n^3) post-processing
(damped Cholesky inverses, eigendecompositions) is sharded BY FACTOR instead of repeated on every
rank: ``partition_by_cost`` bins the factors largest-first (cost ~ n^3: ResNet-18 has 3 x 4608^2,
4 x 2304^2, ...), each rank processes its bin, and the results travel as one packed broadcast per
owner (``sharded_factor_map``; total traffic = one all-gather of the results).  A side effect worth
having: all ranks use bit-identical eigenvector bases (no per-rank sign / ordering differences
before the all-reduce of the corrected eigenvalues).

The reference has no multi-device support (README "future ideas"); this module is new design,
checked by comparing R-rank results with the 1-rank result on identical data.
"""

from __future__ import annotations


from collections.abc import Iterable, Sequence

import torch
import torch.distributed as dist
from torch import Tensor

from curvlinops_amd import _hip
from curvlinops_amd.linop import PyTorchLinearOperator


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_batches(data: Sequence, rank: int | None = None, world_size: int | None = None) -> list:
    """Round-robin split of a list of mini-batches across ranks (rank r takes r, r+R, ...)."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world_size is None:
        world_size = dist.get_world_size() if is_distributed() else 1
    return [b for i, b in enumerate(data) if i % world_size == rank]


def shard_rows(X: Tensor, y: Tensor, rank: int | None = None, world_size: int | None = None) -> tuple[Tensor, Tensor]:
    """Contiguous split of ONE mini-batch along dim 0 (near-equal parts)."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world_size is None:
        world_size = dist.get_world_size() if is_distributed() else 1
    bounds = torch.linspace(0, X.shape[0], world_size + 1).round().long().tolist()
    return X[bounds[rank] : bounds[rank + 1]], y[bounds[rank] : bounds[rank + 1]]


def allreduce_tensors_(tensors: Iterable[Tensor], group=None) -> None:
    """In-place sum over ranks of many tensors with ONE collective on a packed flat buffer
    (few large messages suit xGMI's point-to-point links better than many small ones)."""
    tensors = [t for t in tensors]
    if not tensors or not is_distributed():
        return
    if len(tensors) == 1 and tensors[0].is_contiguous():
        dist.all_reduce(tensors[0], op=dist.ReduceOp.SUM, group=group)
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off : off + n].view_as(t))
        off += n


def allreduce_flat_(flat: Tensor, group=None):
    """In-place sum over ranks of ONE pre-packed contiguous buffer (the factor buffer of a
    data-parallel KFAC build: every factor is a view into it, so nothing is packed or copied back).
    """
    if is_distributed() and flat.numel():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


def partition_by_cost(costs: Sequence[float], world_size: int) -> list[int]:
    """Owner rank of every item: longest-processing-time-first bin packing (deterministic, the same
    on every rank)."""
    load = [0.0] * world_size
    owner = [0] * len(costs)
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world_size), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += costs[i]
    return owner


def sharded_factor_map(mats: Sequence[Tensor], local_fn, out_shapes, group=None) -> list[tuple[Tensor, ...]]:
    """``local_fn(list_of_matrices) -> list of tuples of tensors`` applied to identical (replicated)
    square matrices, each matrix on ONE rank only; every rank returns all results.

    ``out_shapes(n)`` gives the shapes of the result tensors for an ``n x n`` input (needed by the
    ranks that do not compute it).  Results are exchanged with one packed ``broadcast`` per owner."""
    mats = list(mats)
    if not is_distributed() or not mats:
        return local_fn(mats)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    owner = partition_by_cost([float(m.shape[0]) ** 3 for m in mats], world)
    mine = [i for i, r in enumerate(owner) if r == rank]
    local = local_fn([mats[i] for i in mine]) if mine else []
    results: list = [None] * len(mats)
    for i, res in zip(mine, local):
        results[i] = tuple(res)
    ref = mats[0]
    for r in range(world):
        idx = [i for i, o in enumerate(owner) if o == r]
        if not idx:
            continue
        shapes = [[tuple(sh) for sh in out_shapes(mats[i].shape[0])] for i in idx]
        total = sum(_numel(sh) for shs in shapes for sh in shs)
        if r == rank:
            flat = torch.cat([t.reshape(-1).to(ref.dtype) for i in idx for t in results[i]])
        else:
            flat = torch.empty(total, device=ref.device, dtype=ref.dtype)
        src = r if group is None else dist.get_global_rank(group, r)
        dist.broadcast(flat, src=src, group=group)
        if r != rank:
            off = 0
            for i, shs in zip(idx, shapes):
                out = []
                for sh in shs:
                    n = _numel(sh)
                    out.append(flat[off : off + n].view(sh))
                    off += n
                results[i] = tuple(out)
    return results


def _numel(shape: tuple[int, ...]) -> int:
    n = 1
    for s in shape:
        n *= s
    return n


class AllReducedLinearOperator(PyTorchLinearOperator):
    """``sum_ranks op_rank``: each rank holds the operator of its data shard (built with the
    global ``num_data``); a product is the local product followed by one all-reduce."""

    def __init__(self, op: PyTorchLinearOperator, group=None):
        super().__init__(op._in_shape, op._out_shape)
        self._op, self._group = op, group
        self.SELF_ADJOINT = op.SELF_ADJOINT

    def _matmat(self, X: list[Tensor]) -> list[Tensor]:
        out = [o.contiguous() for o in self._op._matmat(X)]
        allreduce_tensors_(out, self._group)
        return out

    def __matmul__(self, X):
        """Flat ``[D]`` / ``[D, K]`` operands: the shard's product through the wrapped operator's own
        ``@`` (its fast paths included), then ONE in-place all-reduce of the flat result -- no
        per-parameter views, no pack / unpack copies around the collective."""
        if isinstance(X, Tensor) and X.dim() in (1, 2):
            Y = self._op @ X
            if not Y.is_contiguous():
                Y = Y.contiguous()
            if is_distributed():
                dist.all_reduce(Y, op=dist.ReduceOp.SUM, group=self._group)
            return Y
        return super().__matmul__(X)

    def matmul_async(self, X: Tensor):
        """``(Y, work)``: the shard's product enqueued on the current stream and its all-reduce
        started asynchronously on the collective's own stream; ``work.wait()`` (stream-side, does
        not block the host) makes ``Y`` the reduced product.  Consecutive independent products --
        probe vectors of a trace estimator, the steps of the benchmark -- overlap the 4 D K-byte
        all-reduce of one product with the kernels of the next."""
        # A collective that is still running holds CUs: a persistent grid that needs every CU of the chip
        # (csrc/mlp_mega.hip) would spin beside it, so a product that is queued while the PREVIOUS collective of this
        # operator has not completed (event query, no host wait) takes the launch chain; once it has completed -- or when
        # the caller never overlaps -- the product runs the same persistent kernel as the single-GPU path.  The choice
        # travels as the `flags` argument of THIS operator's C calls (clo_mlp_ggn_matvec), not as process state.
        op = self._op
        nat = getattr(op, "_native", None) if is_distributed() else None
        prev = getattr(self, "_last_work", None)
        busy = prev is not None and not prev.is_completed()
        self.async_route = "chain" if (nat is not None and busy) else "persistent"
        if nat is not None and busy:   # (per call and per thread: nothing on the shared operator object is mutated)
            with nat.plan.flags_override(_hip.MLP_NO_PERSISTENT):
                Y = op @ X
        else:
            Y = op @ X
        if not Y.is_contiguous():
            Y = Y.contiguous()
        work = dist.all_reduce(Y, op=dist.ReduceOp.SUM, group=self._group, async_op=True) if is_distributed() else None
        self._last_work = work
        return Y, work

    def _adjoint(self) -> "AllReducedLinearOperator":
        return AllReducedLinearOperator(self._op.adjoint(), self._group)

    @property
    def device(self) -> torch.device:
        return self._op.device

    @property
    def dtype(self) -> torch.dtype:
        return self._op.dtype
