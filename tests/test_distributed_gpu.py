"""Two ranks sharing ONE GPU over gloo (RCCL needs one device per rank; the round-end multi-GPU run
covers that): the data-parallel KFAC / EKFAC operators with the factor post-processing sharded by
factor must equal the single-process operators on the same data, through the native kernels."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import curvlinops_amd as C
        from curvlinops_amd import _hip
        from curvlinops_amd.dist import AllReducedLinearOperator, shard_batches

        _hip.load()
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        # factor sizes 301 / 700 / 64 ...: large enough for the worker-thread path of the inverses
        model = nn.Sequential(nn.Linear(300, 700), nn.ReLU(), nn.Linear(700, 64), nn.Tanh(), nn.Linear(64, 10)).to(dev)
        params = dict(model.named_parameters())
        data = [(torch.rand(b, 300, device=dev), torch.randint(0, 10, (b,), device=dev)) for b in (40, 24, 33, 31)]
        N = sum(x.shape[0] for x, _ in data)
        loss = nn.CrossEntropyLoss()
        D = sum(p.numel() for p in params.values())
        v = torch.rand(D, 3, device=dev)
        mine = shard_batches(data)
        failed = []

        def rel(a, b):
            return float((a - b).abs().max() / b.abs().max())

        full = C.GGNLinearOperator(model, loss, params, data, check_deterministic=False)
        local = C.GGNLinearOperator(model, loss, params, mine, num_data=N, check_deterministic=False)
        assert local.uses_native_kernels
        red = AllReducedLinearOperator(local)
        if rel(red @ v, full @ v) > 1e-4:
            failed.append("ggn")
        # overlapped form: back-to-back products, each collective started asynchronously; a product queued behind a
        # collective that is still running takes the launch chain, one behind a completed collective the single-GPU route
        want1 = full @ v[:, 0].contiguous()
        routes = set()
        pending = []
        for _ in range(4):
            Y, work = red.matmul_async(v[:, 0].contiguous())
            routes.add(red.async_route)
            pending.append((Y, work))
        for Y, work in pending:
            work.wait()
            if rel(Y, want1) > 1e-4:
                failed.append("ggn matmul_async")
        torch.cuda.synchronize()
        Y, work = red.matmul_async(v[:, 0].contiguous())    # every earlier collective has completed
        work.wait()
        if red.async_route != "persistent" or rel(Y, want1) > 1e-4:
            failed.append(f"ggn matmul_async after completion: route {red.async_route}")
        if not routes <= {"chain", "persistent"}:
            failed.append(f"routes {routes}")
        for cls in (C.KFACLinearOperator, C.EKFACLinearOperator):
            kw = dict(fisher_type="type-2", check_deterministic=False, separate_weight_and_bias=False)
            K1 = cls(model, loss, params, data, **kw)
            KR = cls(model, loss, params, mine, num_data=N, distributed=True, **kw)
            if rel(KR @ v, K1 @ v) > 1e-4:
                failed.append(f"{cls.__name__} matvec")
            # (factors of rank <= 128: the damping sets the conditioning of the comparison)
            err = rel(KR.inverse(damping=1e-1) @ v, K1.inverse(damping=1e-1) @ v)
            if err > 1e-3:
                failed.append(f"{cls.__name__} inverse {err:.2e}")
        torch.cuda.synchronize()
        ret[rank] = failed
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_sharded_factor_postprocessing():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
        assert p.exitcode == 0
    assert dict(ret) == {0: [], 1: []}


def _worker_captured(rank: int, world: int, port: int, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import curvlinops_amd as C
        from benchmarks.models import ResNetToy, kfac_params
        from curvlinops_amd import _hip, computers
        from curvlinops_amd.dist import shard_batches

        _hip.load()
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = ResNetToy().to(dev).eval()
        params = kfac_params(model)
        # rank 0: batches 0, 2 (12 rows, twice the same shape); rank 1: batches 1, 3 (12 and 7 rows)
        data = [(torch.rand(b, 3, 8, 8, device=dev), torch.randint(0, 5, (b,), device=dev)) for b in (12, 12, 12, 7)]
        N = sum(x.shape[0] for x, _ in data)
        mine = shard_batches(data)
        loss = nn.CrossEntropyLoss()
        failed = []

        def rel(a, b):
            return float((a - b).abs().max() / b.abs().max())

        for cls, fisher in ((C.KFACLinearOperator, "empirical"), (C.KFACLinearOperator, "forward-only"),
                            (C.EKFACLinearOperator, "empirical")):
            kw = dict(fisher_type=fisher, check_deterministic=False, separate_weight_and_bias=False)
            computers._CAPTURE = False
            K1 = cls(model, loss, params, data, **kw)                                   # single process, eager
            KE = cls(model, loss, params, mine, num_data=N, distributed=True, **kw)     # two ranks, eager
            computers._CAPTURE = True
            computers.reset_captured_builds()
            before = computers._CAPTURE_REPLAYS
            ops = [cls(model, loss, params, mine, num_data=N, distributed=True, **kw) for _ in range(3)]
            if computers._CAPTURE_REPLAYS - before < 3:
                failed.append(f"{cls.__name__}/{fisher}: no captured replay ran ({computers._CAPTURE_REPLAYS - before})")
            split = [v.split for v in computers._CAPTURED.values() if isinstance(v, computers._CapturedBatch)]
            if not split or not all(split):
                failed.append(f"{cls.__name__}/{fisher}: distributed captures are not split ({split})")
            v = torch.rand(K1.shape[1], 3, device=dev)
            want = K1 @ v
            for name, op in [("eager", KE)] + [(f"captured{i}", o) for i, o in enumerate(ops)]:
                err = rel(op @ v, want)
                if not err < 1e-4:
                    failed.append(f"{cls.__name__}/{fisher}/{name}: {err:.2e}")
        torch.cuda.synchronize()
        ret[rank] = failed
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_captured_distributed_build():
    """`distributed=True` replays the SAME hipGraphs as the single-GPU build (split behind the last input covariance, the
    all-reduce of the input covariances started between the halves): capture on == eager == single process (1e-4), and a
    captured replay really ran on every rank."""
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_captured, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
        assert p.exitcode == 0
    assert dict(ret) == {0: [], 1: []}
