"""Benchmark of the hot path on MI355X: GGN matvecs/s on the 10M-parameter MLP of
BASELINE.json configs[1] (C2: Linear(1024,2688)-ReLU-Linear(2688,2688)-ReLU-Linear(2688,10),
D = 10 010 122, MSE(mean), fp32, synthetic data).

    python bench.py --gpus N --steps K --warmup W [--batch B]

One "step" = one ``GGNLinearOperator @ v`` over this rank's mini-batch shard (B rows, default 8:
the HBM-bound regime the north star targets) through the public operator API, followed -- when
N > 1 -- by the RCCL all-reduce that sums the per-shard products (started asynchronously: the
collective of step i overlaps the kernels of step i + 1; all K products are fully reduced inside the
timed region).  Weak scaling: every rank holds
B rows, ``num_data = N * B``.  ``value`` = (N * K) shard-matvecs / max-over-ranks time.  The probe
vectors rotate over several buffers so that v and the result stream from / to HBM instead of
living in the 256 MiB Infinity Cache; the weights are constant across products, as in any real
use of the operator.

Extra legs on rank 0 at N = 1 (outside the timed region):
* ``roofline``: the same steps with the library's HIP-event instrumentation on (events on the
  launch stream); the dominant kernel family is the forward+JVP weight stream
  (``fwd_mfma_first_kernel`` + ``fwd_mfma_kernel``: W and V of every layer read once = 8 B per
  parameter of the 12 B/parameter a matvec moves algorithmically); achieved = its algorithmic
  bytes / its summed launch durations, against the 8 TB/s HBM3E peak.
* ``cpu_baseline``: the NumPy oracle (``oracle/mlp_numpy.py``, "port") in float32 on the host
  cores for a bounded number of matvecs of the same workload.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIMS = [1024, 2688, 2688, 10]
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_problem(device, batch: int, seed: int):
    torch.manual_seed(0)
    model = nn.Sequential(
        nn.Linear(DIMS[0], DIMS[1]), nn.ReLU(), nn.Linear(DIMS[1], DIMS[2]), nn.ReLU(), nn.Linear(DIMS[2], DIMS[3])
    ).to(device)
    g = torch.Generator(device="cpu").manual_seed(1234 + seed)  # a different shard per rank
    X = torch.rand(batch, DIMS[0], generator=g).to(device)
    y = torch.rand(batch, DIMS[3], generator=g).to(device)
    return model, X, y


def pmc_traffic_per_launch(kernel: str):
    """Mean HBM bytes per launch of `kernel` from the committed PMC summary (collected by separate
    rocprofv3 --pmc passes of this same command; see tools/pmc_summary.py); None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_c2_n8_pmc_traffic.json")
    try:
        rows = [k for k in json.load(open(path))["kernels"] if kernel in k["kernel"]]
    except (OSError, ValueError, KeyError):
        return None
    if not rows:
        return None
    return sum(k["hbm_bytes"] for k in rows) / len(rows)


def rocprof_avg_us(kernels):
    """Launch-weighted mean duration of `kernels` in the committed rocprofv3 --stats summary."""
    path = os.path.join(ROOT, "profiles", "r01_c2_n8_bench_kernel_stats.txt")
    tot = cnt = 0.0
    try:
        for line in open(path):
            f = line.split(None, 4)
            if len(f) == 5 and not line.startswith("#") and any(k in f[4] for k in kernels):
                cnt += int(f[0]); tot += float(f[1])
    except (OSError, ValueError):
        return None
    return tot / cnt if cnt else None


def other_points(model, params, device, D: int) -> dict:
    """Untimed-region extras on the same network (not the headline): larger mini-batches (MFMA-bound
    regime of SURVEY 8d) and a K = 32 probe block through the K-column kernels."""
    import curvlinops_amd as C

    def us_per_call(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / n

    out = {}
    v = torch.rand(D, device=device)
    for rows in (128, 512):
        X, y = torch.rand(rows, DIMS[0], device=device), torch.rand(rows, DIMS[3], device=device)
        G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
        us = us_per_call(lambda: G @ v, 20)
        out[f"rows{rows}"] = {"us_per_matvec": us, "alg_tflops": 10.0 * rows * D / us / 1e6,
                              "frac_of_f32_mfma_peak": 10.0 * rows * D / us / 1e6 / 157.3}
    X, y = torch.rand(8, DIMS[0], device=device), torch.rand(8, DIMS[3], device=device)
    G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
    V = torch.rand(D, 32, device=device)
    us = us_per_call(lambda: G @ V, 4)
    out["k32_columns_rows8"] = {"us_per_column": us / 32, "columns_per_s": 32e6 / us,
                                "alg_bytes_per_column": 8 * D, "achieved_GBps": 8 * D * 32 / us / 1e3}
    return out


def cpu_baseline(batch: int, budget_s: float = 12.0) -> dict:
    """Time the float32 NumPy oracle on the host cores for a bounded number of matvecs."""
    from oracle import mlp_numpy as O

    rng = np.random.default_rng(0)
    Ws = [(rng.random((DIMS[i + 1], DIMS[i]), dtype=np.float32) - 0.5) / np.sqrt(DIMS[i]) for i in range(3)]
    bs = [rng.random(DIMS[i + 1], dtype=np.float32) - 0.5 for i in range(3)]
    vWs = [rng.random(W.shape, dtype=np.float32) for W in Ws]
    vbs = [rng.random(b.shape, dtype=np.float32) for b in bs]
    X = rng.random((batch, DIMS[0]), dtype=np.float32)
    y = rng.random((batch, DIMS[3]), dtype=np.float32)
    acts = ["relu", "relu", "identity"]
    O.ggn_matvec_batch(Ws, bs, acts, X, y, "mse", "mean", vWs, vbs)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        O.ggn_matvec_batch(Ws, bs, acts, X, y, "mse", "mean", vWs, vbs)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 2000:
            break
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        cores = os.cpu_count() or 1
    return {"value": n / el, "unit": "matvecs/s", "cores": int(cores), "kind": "port",
            "sample": f"{n} float32 GGN matvecs of the same C2 workload (B={batch}) with oracle/mlp_numpy.py in {el:.1f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8, help="mini-batch rows per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / cpu_baseline legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU (the modulo only matters for the single-GPU dry run of the N > 1 code path)
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("CLO_BENCH_BACKEND", "nccl")  # "nccl" = RCCL over xGMI; "gloo" for dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    import curvlinops_amd as C
    from curvlinops_amd import _hip
    from curvlinops_amd.dist import AllReducedLinearOperator

    _hip.load()  # no fallback: the HIP library must be there
    model, X, y = build_problem(device, args.batch, seed=rank)
    params = dict(model.named_parameters())
    D = sum(p.numel() for p in params.values())
    G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False,
                            num_data=args.batch * world)
    assert G.uses_native_kernels, "C2 must run on the native HIP path"
    op = AllReducedLinearOperator(G) if world > 1 else G

    nbuf = 8  # 8 x 40 MB probes + 8 results > 256 MiB Infinity Cache
    g = torch.Generator(device="cpu").manual_seed(7)
    vs = [torch.rand(D, generator=g).to(device) for _ in range(nbuf)]

    # N > 1: the 40 MB all-reduce of product i runs on RCCL's stream while the kernels of product
    # i + 1 run on the compute stream (at most two collectives in flight; every product of the K
    # timed steps is fully reduced before the closing synchronisation).  CLO_BENCH_OVERLAP=0: strictly
    # sequential product -> all-reduce.
    overlap = world > 1 and os.environ.get("CLO_BENCH_OVERLAP", "1") != "0"
    inflight: list = []

    def step(i: int):
        if not overlap:
            return op @ vs[i % nbuf]
        y, work = op.matmul_async(vs[i % nbuf])
        inflight.append((y, work))
        if len(inflight) > 2:
            inflight.pop(0)[1].wait()
        return y

    def drain():
        while inflight:
            inflight.pop(0)[1].wait()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    sync()
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        out = step(i)
    drain()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out is not None and torch.isfinite(out).all()

    ms_per_step = 1e3 * elapsed / args.steps
    result = {
        "metric": "curvature matvecs/s (GGN, D=10M MLP)",
        "value": world * args.steps / elapsed,
        "unit": "matvecs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "C2: GGNLinearOperator @ v, MLP 1024-2688-2688-10 (D=10010122), MSE mean, "
                        f"{args.batch} rows per GPU, K=1",
            "rows_per_gpu": args.batch,
            "parallelism": (f"dp{world} (data shards + RCCL all-reduce of the [D] result"
                            + (", overlapped with the next product)" if overlap else ")")) if world > 1 else "single GPU",
        },
    }

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- roofline leg: same steps with per-kernel HIP events on the launch stream
        _hip.prof_enable(True)
        nprof = min(args.steps, 200)
        for i in range(nprof):
            step(i)
        torch.cuda.synchronize()
        prof = _hip.prof_collect()
        _hip.prof_enable(False)
        # dominant kernel family: the forward+JVP weight stream (fwd_mfma_first_kernel for layer 1,
        # fwd_mfma_kernel for the others): reads W and V of every layer exactly once = 8 B/parameter
        fam = max(prof, key=lambda k: prof[k]["ms"])
        r = prof[fam]
        kernel_names = {"fwd_mfma": ["fwd_mfma_first_kernel", "fwd_mfma_kernel"], "bwd_dprev": ["bwd_fused_kernel"],
                        "outer_all": ["outer_all_kernel"]}.get(fam, [fam])
        tr = [pmc_traffic_per_launch(k) for k in kernel_names]
        traffic = (sum(t for t in tr if t) / max(sum(1 for t in tr if t), 1)) if any(tr) else None
        achieved = r["alg_bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0.0
        kernels_ms = sum(v["ms"] for v in prof.values()) / nprof
        result["roofline"] = {
            "bound": "hbm",
            "kernel": " + ".join(kernel_names),
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": "profiles/r01_c2_n8_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                              "separate passes, FETCH doubled per the gfx950 note; mean over the family's launches)" if traffic else None,
            "launches": r["launches"],
            "avg_launch_us": 1e3 * r["ms"] / max(r["launches"], 1),
            "avg_launch_us_note": "HIP-event interval around each launch on the launch stream: includes the "
                                  "dispatch latency (~2.5 us) that rocprofv3's kernel durations exclude",
            "rocprof_avg_kernel_us": rocprof_avg_us(kernel_names),
            "alg_bytes_per_launch": r["alg_bytes"] / max(r["launches"], 1),
            "whole_matvec": {
                "alg_bytes": 12 * D,
                "achieved_GBps": 12 * D / (ms_per_step * 1e-3) / 1e9,
                "frac": 12 * D / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "kernel_ms_per_matvec": kernels_ms,
            },
            "per_family_ms_per_matvec": {k: v["ms"] / nprof for k, v in prof.items()},
        }
        result["cpu_baseline"] = cpu_baseline(args.batch)
        result["other_points"] = other_points(model, params, device, D)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
