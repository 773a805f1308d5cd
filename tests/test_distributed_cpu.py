"""world_size-2 gloo tests of the data-parallel path on CPU: R-rank results (local shard +
packed all-reduce) must equal the single-process result on the same data."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(failed: list, label: int, cond) -> bool:
    if not cond:
        failed.append(label)
    return bool(cond)


def _worker(rank: int, world: int, port: int, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import curvlinops_amd as C
        from curvlinops_amd.dist import AllReducedLinearOperator, shard_batches, shard_rows

        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(6, 7), nn.Tanh(), nn.Linear(7, 3)).double()
        params = dict(model.named_parameters())
        data = [(torch.rand(b, 6, dtype=torch.float64), torch.randint(0, 3, (b,))) for b in (5, 3, 8, 4)]
        N = sum(x.shape[0] for x, _ in data)
        loss = nn.CrossEntropyLoss()
        v = torch.rand(sum(p.numel() for p in params.values()), 2, dtype=torch.float64)

        full = C.GGNLinearOperator(model, loss, params, data)
        mine = shard_batches(data)
        local = C.GGNLinearOperator(model, loss, params, mine, num_data=N, check_deterministic=False)
        failed: list = []
        ok = _check(failed, 0, torch.allclose(AllReducedLinearOperator(local) @ v, full @ v, rtol=1e-10, atol=1e-12))

        # 1-D vector and tensor-list formats take the same collective
        AR = AllReducedLinearOperator(local)
        ok &= _check(failed, 1, torch.allclose(AR @ v[:, 0].contiguous(), (full @ v)[:, 0], rtol=1e-10, atol=1e-12))
        shapes = [p.shape for p in params.values()]
        vl = [c.reshape(s) for c, s in zip(v[:, 1].split([s.numel() for s in shapes]), shapes)]
        got = torch.cat([o.flatten() for o in AR @ vl])
        ok &= _check(failed, 2, torch.allclose(got, (full @ v)[:, 1], rtol=1e-10, atol=1e-12))

        # asynchronous form: product now, collective in flight, reduced after wait()
        Ya, work = AR.matmul_async(v)
        work.wait()
        ok &= _check(failed, 50, torch.allclose(Ya, full @ v, rtol=1e-10, atol=1e-12))

        # one mini-batch split by rows
        X, y = torch.cat([x for x, _ in data]), torch.cat([t for _, t in data])
        Xr, yr = shard_rows(X, y)
        local = C.HessianLinearOperator(model, loss, params, [(Xr, yr)], num_data=N, check_deterministic=False)
        fullH = C.HessianLinearOperator(model, loss, params, [(X, y)])
        ok &= _check(failed, 3, torch.allclose(AllReducedLinearOperator(local) @ v, fullH @ v, rtol=1e-10, atol=1e-12))

        for cls in (C.KFACLinearOperator, C.EKFACLinearOperator):
            K1 = cls(model, loss, params, data, fisher_type="type-2", check_deterministic=False)
            KR = cls(model, loss, params, mine, fisher_type="type-2", num_data=N, check_deterministic=False,
                     distributed=True)
            ok &= _check(failed, 4, torch.allclose(KR @ v, K1 @ v, rtol=1e-8, atol=1e-10))
            # damped inverses: the Cholesky inverses (KFAC) are sharded by factor over the ranks and
            # exchanged with packed broadcasts, the EKFAC bases come from the sharded eigh
            ok &= _check(failed, 5, torch.allclose(KR.inverse(damping=1e-2) @ v, K1.inverse(damping=1e-2) @ v, rtol=1e-8, atol=1e-10))

        # a rank without any mini-batch still takes part in both factor collectives (zero contribution)
        lopsided = data if rank == 0 else []
        K1 = C.KFACLinearOperator(model, loss, params, data, fisher_type="empirical", check_deterministic=False,
                                  separate_weight_and_bias=False)
        KR = C.KFACLinearOperator(model, loss, params, lopsided, fisher_type="empirical", num_data=N,
                                  num_per_example_loss_terms=1, check_deterministic=False,
                                  separate_weight_and_bias=False, distributed=True)
        ok &= _check(failed, 9, torch.allclose(KR @ v, K1 @ v, rtol=1e-8, atol=1e-10))

        # the sharding helpers themselves
        from curvlinops_amd import linalg_native
        from curvlinops_amd.dist import partition_by_cost, sharded_factor_map

        owner = partition_by_cost([4.0**3, 9.0**3, 2.0**3, 9.0**3, 5.0**3], 2)
        ok &= _check(failed, 6, owner == [1, 0, 1, 1, 0])  # largest first onto the emptier bin
        g = torch.Generator().manual_seed(1)
        mats = []
        for n in (4, 9, 2, 9, 5):
            Z = torch.rand(n + 3, n, generator=g, dtype=torch.float64)
            mats.append(Z.T @ Z)
        both = sharded_factor_map(mats, linalg_native.eigh_many, lambda n: [(n,), (n, n)])
        for M, (lam, Q) in zip(mats, both):
            ok &= _check(failed, 7, torch.allclose(Q @ torch.diag(lam) @ Q.T, M, rtol=1e-10, atol=1e-12))
        with linalg_native.concurrent_inverses(distributed=True):
            invs = [linalg_native.damped_cholesky_inverse(M, 0.1) for M in mats]
        for M, Mi in zip(mats, invs):
            ok &= _check(failed, 8, torch.allclose(Mi, torch.linalg.inv(M + 0.1 * torch.eye(M.shape[0], dtype=M.dtype)), rtol=1e-9, atol=1e-11))
        # a non-positive-definite factor without retry raises on EVERY rank (no rank left waiting)
        bad = torch.eye(3, dtype=torch.float64)
        bad[0, 0] = -1.0
        try:
            with linalg_native.concurrent_inverses(distributed=True):
                for M in [*mats, bad]:
                    linalg_native.damped_cholesky_inverse(M, 0.0, retry_double_precision=False)
            ok = _check(failed, 99, False)
        except RuntimeError:
            pass
        ret[rank] = True if ok else f"failed checks {failed}"
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}
