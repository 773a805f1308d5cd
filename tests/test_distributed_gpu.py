"""Two ranks sharing ONE GPU over gloo (RCCL needs one device per rank; the round-end multi-GPU run
covers that): the data-parallel KFAC / EKFAC operators with the factor post-processing sharded by
factor must equal the single-process operators on the same data, through the native kernels."""

import os
import socket

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
        if rel(AllReducedLinearOperator(local) @ v, full @ v) > 1e-4:
            failed.append("ggn")
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
