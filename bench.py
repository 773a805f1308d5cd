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

Extra legs (outside the timed region of the headline metric, same JSON line):
* ``roofline`` (rank 0, N = 1): ``frac`` is the WHOLE matvec against the HBM roofline --
  ``12 D`` algorithmic bytes (theta, v and the result once each) / the step time of the
  timed region / 8 TB/s -- the quantity the north star's ">= 50 %" refers to.  The per-kernel
  picture (HIP events on the launch stream, live) sits under ``kernels``; ``traffic`` is the HBM
  bytes of one matvec from the committed PMC passes.
* ``kfac`` (every N): the second half of BASELINE.json's metric -- KFAC factor build ms per batch
  on config C4 (ResNet-18, 512 rows per GPU, joint W+b, one MC sample; reference phases
  ``benchmark_utils.py:139-143``, protocol ``benchmark_execute.py:288-301``: min of 5 after one
  warm-up).  N > 1: every rank builds on its shard and the factors are summed by ONE in-place
  all-reduce of the flat factor buffer (weak scaling, 512 rows per rank).
* ``scalable_points`` (N > 1): the same operator at 512 rows per rank (MFMA-bound; the regime in
  which a 40 MB all-reduce per product can hide behind the kernels).
* ``cpu_baseline`` (rank 0, N = 1): this package's own operators on CPU tensors (torch ops, fp32,
  all physical cores, min of 10 after one warm-up -- the reference's protocol), GGN matvec of the
  same workload and a bounded KFAC factor build.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Round 6: the bench no longer sets GPU_MAX_HW_QUEUES (rounds 3-5 ran with 16): every figure is what a drop-in user of the
# package gets with the runtime's default of 4 hardware queues.  What 16 queues change (profiles/r06_kfac_capture_rule.txt):
# the twelve-stream eigensolver leg 101 -> 78 ms, the captured KFAC build 4.25 -> 5.06 ms (the package then captures a
# one-branch graph, computers._CAPTURE_BRANCHES), the 42 Cholesky inverses 10.9 vs 11.2 ms, the headline matvec nothing.

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


def latest_profile(suffix: str):
    """Path of the newest committed ``profiles/rNN_<suffix>`` (highest round), or None."""
    import glob

    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return hits[-1] if hits else None


def csrc_sha16() -> str:
    """Fingerprint of the kernel sources: committed profile summaries carry the one they were collected with, so a
    figure read from ``profiles/`` can be flagged when the kernels have changed since."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "curvlinops_amd", "csrc", "*.hip"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(kernel: str):
    """Mean HBM bytes per launch of `kernel` from the committed PMC summary (collected by separate
    rocprofv3 --pmc passes of this same command; see tools/pmc_summary.py); None if absent."""
    path = latest_profile("c2_n8_pmc_traffic.json")
    try:
        rows = [k for k in json.load(open(path))["kernels"] if kernel in k["kernel"]]
    except (OSError, ValueError, KeyError, TypeError):
        return None
    if not rows:
        return None
    return sum(k["hbm_bytes"] for k in rows) / len(rows)


def pmc_traffic_meta():
    path = latest_profile("c2_n8_pmc_traffic.json")
    try:
        sha = json.load(open(path)).get("csrc_sha16")
    except (OSError, ValueError, TypeError):
        return None, None
    return os.path.relpath(path, ROOT), (None if sha is None else sha != csrc_sha16())


def rocprof_avg_us(kernels):
    """Launch-weighted mean duration of `kernels` in the committed rocprofv3 --stats summary."""
    path = latest_profile("c2_n8_bench_kernel_stats.txt") or ""
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
        for _ in range(3):   # (the first calls of a shape allocate workspaces and run at a ramping clock)
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
        us = us_per_call(lambda: G @ v, 30)
        out[f"rows{rows}"] = {"us_per_matvec": us, "alg_tflops": 10.0 * rows * D / us / 1e6,
                              "frac_of_f32_mfma_peak": 10.0 * rows * D / us / 1e6 / 157.3}
    X, y = torch.rand(8, DIMS[0], device=device), torch.rand(8, DIMS[3], device=device)
    G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
    for K in (32, 64):
        Vs = [torch.rand(D, K, device=device) for _ in range(2)]   # two blocks: V and the result stream from / to HBM
        state = {"i": 0}

        def prod():
            state["i"] ^= 1
            return G @ Vs[state["i"]]

        us = us_per_call(prod, 6)
        out[f"k{K}_columns_rows8"] = {"us_per_column": us / K, "columns_per_s": K * 1e6 / us,
                                      "alg_bytes_per_column": 8 * D, "achieved_GBps": 8 * D * K / us / 1e3,
                                      "frac_of_hbm_peak": 8 * D * K / us / 1e3 / HBM_PEAK_GBPS}
        del Vs
    return out


def physical_cores() -> int:
    try:
        import psutil

        return int(psutil.cpu_count(logical=False) or os.cpu_count() or 1)
    except Exception:  # noqa: BLE001
        return os.cpu_count() or 1


def cpu_baseline(batch: int) -> dict:
    """This package's operators on CPU tensors (the torch.func path of `curvlinops_amd.curvature`,
    validated against the reference goldens by the CPU test-suite), fp32, the reference's protocol:
    `perf_counter`, min of the repeats after one warm-up
    (`docs/examples/basic_usage/benchmark_execute.py:288-301`).  The thread count is SWEPT (8, 16, 32, 64, all
    physical cores; the shared host oversubscribes easily) and the best setting is the reported baseline."""
    import curvlinops_amd as C

    cores = physical_cores()
    cpu = torch.device("cpu")
    model, X, y = build_problem(cpu, batch, seed=0)
    params = dict(model.named_parameters())
    D = sum(p.numel() for p in params.values())
    G = C.GGNLinearOperator(model, nn.MSELoss(), params, [(X, y)], check_deterministic=False)
    v = torch.rand(D)
    t_all = time.perf_counter()
    sweep = {}
    for nt in sorted({t for t in (8, 16, 32, 64, cores) if t <= cores}):
        torch.set_num_threads(nt)
        G @ v  # warm-up
        best = float("inf")
        for _ in range(6):
            t0 = time.perf_counter()
            G @ v
            best = min(best, time.perf_counter() - t0)
        sweep[nt] = 1.0 / best
    best_nt = max(sweep, key=sweep.get)
    out = {"value": sweep[best_nt], "unit": "matvecs/s", "cores": best_nt, "kind": "port",
           "thread_sweep_matvecs_per_s": {str(k): v_ for k, v_ in sweep.items()}, "physical_cores": cores,
           "sample": f"float32 GGN matvecs of the same C2 workload (B={batch}) with curvlinops_amd.GGNLinearOperator on "
                     f"CPU tensors: min of 6 after 1 warm-up at each of {sorted(sweep)} threads, best = {best_nt} "
                     f"threads; {time.perf_counter() - t_all:.1f} s of CPU work"}
    # bounded KFAC sample: ResNet-18 factor build on 128 rows at the best thread count (the GPU leg uses 512 rows
    # per GPU: compare per row)
    try:
        from benchmarks.models import ResNet18, kfac_params

        torch.set_num_threads(best_nt)
        torch.manual_seed(0)
        net = ResNet18().eval()
        kp = kfac_params(net)
        rows = 128
        Xc, yc = torch.rand(rows, 3, 32, 32), torch.randint(0, 10, (rows,))
        kw = dict(fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=rows)
        t0 = time.perf_counter()
        C.KFACLinearOperator(net, nn.CrossEntropyLoss(), kp, [(Xc, yc)], **kw)
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        C.KFACLinearOperator(net, nn.CrossEntropyLoss(), kp, [(Xc, yc)], **kw)
        second = time.perf_counter() - t0
        ms = 1e3 * min(first, second)
        out["kfac_factor_build"] = {"ms_per_batch": ms, "rows": rows, "ms_per_row": ms / rows, "cores": best_nt,
                                    "sample": f"ResNet-18 KFAC factor build, {rows} rows, min of 2, CPU tensors, "
                                              f"{best_nt} threads"}
    except Exception as e:  # noqa: BLE001
        out["kfac_factor_build"] = {"error": repr(e)}
    return out


def secondary_configs(device) -> dict:
    """Driver-visible numbers for the remaining BASELINE configs (rank 0, N = 1, outside the timed region):
    EKFAC phases on C4 (reference phases `benchmark_utils.py:139-143`) and C5 = 12-layer d = 768 encoder,
    `EFLinearOperator` matvec + `hutchpp_trace` with 96 products (three K = 32 probe blocks)."""
    import curvlinops_amd as C
    from benchmarks.models import Encoder, ResNet18, kfac_params
    from curvlinops_amd import linalg_native

    def timed(fn, repeats):
        fn()
        torch.cuda.synchronize()
        best, out = float("inf"), None
        for _ in range(repeats):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return 1e3 * best, out

    def median_of(fn, repeats):
        """(median ms, [min, max] ms, last result) after one warm-up: the protocol of the noisy multi-thread legs."""
        fn()
        torch.cuda.synchronize()
        ts, out = [], None
        for _ in range(repeats):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        ts.sort()
        return ts[len(ts) // 2], [ts[0], ts[-1]], out

    out = {}
    skip = os.environ.get("CLO_BENCH_SKIP", "").split(",")   # diagnosis only: leave out c3 / c4 / c5
    if "c3" not in skip:
        # ---- C3: LeNet-5 on 32 x 32 inputs, B = 1024, KFAC factor build + damped inverse + matvec (BASELINE configs[2])
        from benchmarks.models import lenet5

        torch.manual_seed(0)
        net3 = lenet5().to(device)
        p3 = dict(net3.named_parameters())
        X3, y3 = torch.rand(1024, 1, 32, 32, device=device), torch.randint(0, 10, (1024,), device=device)
        kw3 = dict(separate_weight_and_bias=False, check_deterministic=False, num_data=1024)
        c3 = {"rows": 1024}
        for ft in ("mc", "type-2"):
            if "c3" + ft in skip:
                continue
            ms, sp, K3 = median_of(lambda: C.KFACLinearOperator(net3, nn.CrossEntropyLoss(), p3, [(X3, y3)], fisher_type=ft, **kw3), 5)
            c3[f"factor_build_ms_{ft}"] = ms
            c3[f"factor_build_ms_{ft}_min_max"] = sp
        K3inv = None
        if "c3inv" not in skip:
            c3["inverse_ms"], c3["inverse_ms_min_max"], K3inv = median_of(lambda: K3.inverse(damping=1e-3), 5)
        v3 = torch.rand(K3.shape[1], device=device)
        if "c3mv" not in skip:
            c3["matvec_ms"], _, _ = median_of(lambda: K3 @ v3, 5)
            if K3inv is not None:
                c3["inverse_matvec_ms"], _, _ = median_of(lambda: K3inv @ v3, 5)
        c3["note"] = "LeNet-5, joint W+b, CE mean; medians of 5 after one warm-up; type-2 = ten backpropagated vectors in one batched pass"
        out["c3_kfac_lenet5"] = c3
        del net3, p3, K3, K3inv
    if "c4" not in skip:
        torch.manual_seed(0)
        model = ResNet18().to(device).eval()
        params = kfac_params(model)
        B = 512
        X, y = torch.rand(B, 3, 32, 32, device=device), torch.randint(0, 10, (B,), device=device)
        kw = dict(fisher_type="mc", separate_weight_and_bias=False, check_deterministic=False, num_data=B)
        K = C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)
        facs = [S for blk in K[1] for S in blk]
        ek = {"rows": B}
        ek["eigh_ms"], ek["eigh_ms_min_max"], _ = median_of(lambda: linalg_native.eigh_many(facs), 5)   # six host threads: noisy
        ek["ekfac_total_ms"], ek["ekfac_total_ms_min_max"], E = median_of(
            lambda: C.EKFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw), 5)
        v = torch.rand(E.shape[1], device=device)
        ek["ekfac_matvec_ms"], _ = timed(lambda: E @ v, 3)
        ek["note"] = ("EKFAC = factors + eigendecompositions of the 42 factors (dead-feature rows deflated, normalised, "
                      "hand-written solver end to end: Householder reduction, tridiagonal divide & conquer batched per "
                      "factor size, block-reflector back-transformation; 6 worker streams, units sized by a measured "
                      "wall-time model; orthogonality / residual verified, float64 retry) + eigenvalue-correction sweep; "
                      "eigh_ms / ekfac_total_ms = MEDIAN of 5 calls after one warm-up, [min, max] beside them")
        ek["eigh_policy"] = "native (clo_sytrd_f32 persistent panels -> divide & conquer -> block reflectors)"
        ek["hip_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default)")
        if os.environ.get("CLO_BENCH_EKFAC_16Q", "1") == "1" and "GPU_MAX_HW_QUEUES" not in os.environ:
            # the same leg in a child process with 16 hardware queues (the runtime reads the variable at start-up): the
            # twelve eigensolver streams then get a queue each -- the setting rounds 3-5 benchmarked with
            import subprocess

            try:
                env = dict(os.environ, GPU_MAX_HW_QUEUES="16", CLO_BENCH_SKIP="c3,c5", CLO_BENCH_EKFAC_16Q="0")
                res = subprocess.run([sys.executable, os.path.abspath(__file__), "--secondary-only"], env=env,
                                     capture_output=True, text=True, timeout=600)
                sub = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])["c4_ekfac_resnet18"]
                ek["with_16_hw_queues"] = {k: sub[k] for k in ("eigh_ms", "eigh_ms_min_max", "ekfac_total_ms",
                                                                "ekfac_total_ms_min_max", "ekfac_matvec_ms")}
            except Exception as e:  # noqa: BLE001
                ek["with_16_hw_queues"] = {"error": repr(e)}
        out["c4_ekfac_resnet18"] = ek
        del K, E, facs, model, params
        torch.cuda.empty_cache()
    if "c5" not in skip:
        torch.manual_seed(0)
        enc = Encoder().to(device).eval()
        p5 = dict(enc.named_parameters())
        X5, y5 = torch.rand(8, 128, 768, device=device), torch.randint(0, 10, (8,), device=device)
        EF = C.EFLinearOperator(enc, nn.CrossEntropyLoss(), p5, [(X5, y5)], check_deterministic=False, num_data=8)
        D5 = EF.shape[1]
        v5 = torch.rand(D5, device=device)
        c5 = {"D": D5, "rows": 8, "seq_len": 128}
        c5["ef_matvec_ms"], c5["ef_matvec_ms_min_max"], _ = median_of(lambda: EF @ v5, 5)
        t0 = time.perf_counter()
        C.hutchpp_trace(EF, num_matvecs=96)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr = C.hutchpp_trace(EF, num_matvecs=96)
        torch.cuda.synchronize()
        c5["hutchpp_96_ms"] = 1e3 * min(t1 - t0, time.perf_counter() - t1)
        c5["hutchpp_trace"] = float(tr)
        c5["note"] = "general net: torch.func products on the GPU; probes, Gram-route range basis and reductions native"
        out["c5_encoder_ef_hutchpp"] = c5
    return out


# --------------------------------------------------------------------------------------------
# KFAC factor build (BASELINE config C4): the second half of the metric
# --------------------------------------------------------------------------------------------
MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md


def kfac_clo_kernel_us():
    """Sum of the clo:: kernel durations of one warm factor build in the committed rocprofv3 summary."""
    try:
        for line in open(latest_profile("kfac_resnet18_build_kernels.txt") or ""):
            if line.startswith("clo_kernel_us"):
                return float(line.split()[1])
    except (OSError, ValueError):
        pass
    return None


def kfac_leg(device, world: int, rank: int, rows: int = 512, repeats: int = 5) -> dict:
    import curvlinops_amd as C
    from benchmarks.models import ResNet18, kfac_params

    torch.manual_seed(0)
    model = ResNet18().to(device).eval()
    params = kfac_params(model)
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    X = torch.rand(rows, 3, 32, 32, generator=g).to(device)
    y = torch.randint(0, 10, (rows,), generator=g).to(device)
    kw = dict(fisher_type="mc", mc_samples=1, separate_weight_and_bias=False, check_deterministic=False,
              num_data=rows * world, distributed=world > 1)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def build():
        return C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw)

    K = build()  # warm-up (first-seen shape: eager)
    K = build()  # (the configuration comes back: captured)
    best = float("inf")
    build_times = []
    for _ in range(repeats):
        sync()
        t0 = time.perf_counter()
        K = build()
        sync()
        build_times.append(time.perf_counter() - t0)
        best = min(best, build_times[-1])
    if world > 1:
        t = torch.tensor([best], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        best = float(t.item())
    from curvlinops_amd import computers as _computers

    captured_graphs = [v for v in _computers._CAPTURED.values() if isinstance(v, _computers._CapturedBatch)]
    parts = {}
    if world > 1:
        # what the N-rank figure is made of: the SAME shard built without any collective (the single-GPU code path and
        # graphs), and the all-reduce of a buffer of the factors' size on its own
        kw1 = dict(kw, distributed=False)
        C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw1)
        C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw1)
        t_local = float("inf")
        for _ in range(repeats):
            sync()
            t0 = time.perf_counter()
            C.KFACLinearOperator(model, nn.CrossEntropyLoss(), params, [(X, y)], **kw1)
            torch.cuda.synchronize()
            t_local = min(t_local, time.perf_counter() - t0)
        nfl = sum(f.numel() for blk in K[1] for f in blk)
        buf = torch.zeros(nfl, device=device)
        dist.all_reduce(buf)
        t_ar = float("inf")
        for _ in range(repeats):
            sync()
            t0 = time.perf_counter()
            dist.all_reduce(buf)
            torch.cuda.synchronize()
            t_ar = min(t_ar, time.perf_counter() - t0)
        tt = torch.tensor([t_local, t_ar], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        parts = {"local_build_ms_no_collective": 1e3 * float(tt[0]), "factor_allreduce_alone_ms": 1e3 * float(tt[1]),
                 "factor_allreduce_MB": 4.0 * nfl / 1e6,
                 "note": "ms_per_batch = split-graph replay with the input-covariance all-reduce started between the "
                         "forward and the backward half; local_build = the single-GPU graph on the same shard"}
    # algorithmic work: sum_l 2 B S_l (d_in'^2 + V d_out^2) flop with V = 1 MC sample
    pos = {}

    def shared_positions(mod, o):
        feat = o.shape[1] if isinstance(mod, nn.Conv2d) else o.shape[-1]
        return o.numel() // (o.shape[0] * feat)

    pix = {}   # conv layers whose input covariance comes from the pixel Gram (csrc/conv.hip): order of that Gram

    def note(mod, i, o):
        from curvlinops_amd import computers

        pos[mod] = shared_positions(mod, o)
        x = i[0]
        if isinstance(mod, nn.Conv2d) and x.dim() == 4 and computers._use_pixel_gram(computers._conv_hyperparams(mod), x):
            pix[mod] = (x.shape[1] // mod.groups) * x.shape[2] * x.shape[3]

    hooks = [m.register_forward_hook(note) for m in model.modules() if isinstance(m, (nn.Conv2d, nn.Linear))]
    with torch.no_grad():
        model(X[:2])
    for h in hooks:
        h.remove()
    flops = 0.0
    factor_floats = 0
    for m, S in pos.items():
        d_out = m.weight.shape[0]
        d_in = m.weight[0].numel() + (1 if m.bias is not None else 0)
        flops += 2.0 * rows * S * (d_in**2 + d_out**2)
        factor_floats += d_in**2 + d_out**2
    out = {
        "metric": "KFAC factor build ms/batch (ResNet-18, C4)",
        "ms_per_batch": 1e3 * best,
        "ms_per_batch_median_min_max": [1e3 * sorted(build_times)[len(build_times) // 2], 1e3 * min(build_times),
                                        1e3 * max(build_times)],
        "rows_per_gpu": rows,
        "global_batch": rows * world,
        "n_gpus": world,
        "rows_per_s": rows * world / best,
        "config": "ResNet-18 (torchvision topology, 10 classes, eval), 3x32x32, CE mean, fisher mc x1, joint W+b, "
                  "Linear/Conv2d parameters only" + (", sharded build + ONE all-reduce of the flat factor buffer"
                                                     if world > 1 else ""),
        "protocol": f"min of {repeats} after 2 warm-up builds, device (and ranks) synchronised around each build",
        "route": {"captured_graphs": len(captured_graphs), "split": [g.split for g in captured_graphs],
                  "branches": _computers._CAPTURE_BRANCHES, "graph_replays": _computers._CAPTURE_REPLAYS},
        **({"parts": parts} if parts else {}),
        "factor_gflop_per_gpu": flops / 1e9,
        "factor_buffer_MB": 4.0 * factor_floats / 1e6,
        # (filled below at N = 1: the factor KERNELS' own figure first -- executed flops over their rocprofv3 time --, then
        # SURVEY 8d's full algorithmic figure over the whole build, autograd included)
        "roofline": {"bound": "mfma", "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "whole_build_survey_figure": {
                         "achieved": flops / best / 1e12, "frac": flops / best / 1e12 / MFMA_F32_PEAK_TFLOPS,
                         "note": "SURVEY 8d's factor SYRK flops (patch products, full d x d, symmetry not discounted: NOT "
                                 "what the pixel-Gram route executes) over the WHOLE build time, which also contains the "
                                 "autograd forward / backward pass (host framework)"}},
    }
    if world == 1:
        def fwdbwd():
            o = model(X)
            return torch.autograd.grad(nn.functional.cross_entropy(o, y), list(params.values()))

        fwdbwd()
        t_ag = float("inf")
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fwdbwd()
            torch.cuda.synchronize()
            t_ag = min(t_ag, time.perf_counter() - t0)
        out["gradient_and_loss_ms"] = 1e3 * t_ag
        # the factor kernels themselves: sum of the clo:: kernel durations of ONE warm build from the committed
        # rocprofv3 trace (tools/run_prof_kfac_build.sh), against the flops the SYRKs EXECUTE (upper block
        # triangle of 128-wide tiles; the figure above counts the full d x d products)
        clo_us = kfac_clo_kernel_us()
        if clo_us:
            executed = 0.0
            for m, S in pos.items():
                for which, d in (("a", m.weight[0].numel() + (1 if m.bias is not None else 0)), ("g", m.weight.shape[0])):
                    k = rows * S
                    if which == "a" and m in pix:   # X^T X of X = x as [rows, C H W]: K = rows, order C H W
                        d, k = pix[m], rows
                    nt = max(1, -(-d // 128))
                    executed += 2.0 * k * d * d * (nt + 1) / (2.0 * nt)
            out["roofline"]["achieved"] = executed / clo_us / 1e6
            out["roofline"]["frac"] = executed / clo_us / 1e6 / MFMA_F32_PEAK_TFLOPS
            out["roofline"]["clo_kernels"] = {
                "kernel_ms_rocprof": clo_us / 1e3, "executed_gflop": executed / 1e9,
                "achieved_tflops_executed": executed / clo_us / 1e6,
                "frac_of_f32_mfma_peak": executed / clo_us / 1e6 / MFMA_F32_PEAK_TFLOPS,
                "source": "profiles/" + os.path.basename(latest_profile("kfac_resnet18_build_kernels.txt") or "")
                          + " (rocprofv3 --kernel-trace of one warm build = one replay of the captured hipGraph: pixel-Gram "
                          "SYRKs + fold, the grouped gradient-covariance launch, SYRK / Gram kernels)"}
        v = torch.rand(K.shape[1], device=device)
        K @ v
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            K @ v
        torch.cuda.synchronize()
        out["kfac_matvec_ms"] = 1e3 * (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        Kinv = K.inverse(damping=1e-3)   # first call: workspaces, helper streams / events of the pipelined inverse
        torch.cuda.synchronize()
        out["cholesky_inverse_ms_first_call"] = 1e3 * (time.perf_counter() - t0)
        times = []
        for _ in range(4):
            t0 = time.perf_counter()
            Kinv = K.inverse(damping=1e-3)
            torch.cuda.synchronize()
            times.append(1e3 * (time.perf_counter() - t0))
        out["cholesky_inverse_ms_second_call"] = times[0]
        out["cholesky_inverse_ms_mean_of_4"] = sum(times) / len(times)
        out["cholesky_inverse_ms_median_min_max"] = [sorted(times)[len(times) // 2], min(times), max(times)]
        del Kinv
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this script as N ranks of one node under
    ``torch.distributed.run`` (rendezvous on 127.0.0.1, a free port), one rank per GPU.  Fewer visible
    devices than ranks is an error unless the gloo dry-run backend was asked for."""
    import socket
    import subprocess

    ndev = torch.cuda.device_count()
    if ndev < n and os.environ.get("CLO_BENCH_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"bench.py: --gpus {n} requested but only {ndev} device(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8, help="mini-batch rows per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / cpu_baseline legs")
    ap.add_argument("--secondary-only", action="store_true",
                    help="print only the secondary configurations (C3 / C4-EKFAC / C5; CLO_BENCH_SKIP leaves some out)")
    args = ap.parse_args()
    if args.secondary_only:
        torch.cuda.set_device(0)
        print(json.dumps(secondary_configs(torch.device("cuda", 0))))
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run, RCCL); rank 0 of the children prints the JSON line
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per requested GPU")
    if world > 1 and os.environ.get("CLO_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} device(s) visible")
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
        try:
            return step_once(i)
        except RuntimeError as e:   # (see `timed` below: only the shared-GPU dry run gets here)
            if "timed out inside the persistent kernel" not in str(e):
                raise
            return step_once(i)

    def step_once(i: int):
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
    n_gt1 = None
    if world > 1:
        # the same K steps (a) strictly sequential -- persistent kernel, then a blocking all-reduce --, (b) without any
        # collective (the shard product alone: the single-GPU kernel on this rank)
        def timed(fn0):
            def fn(i):
                # (two ranks on ONE GPU -- the gloo dry run -- cannot keep two persistent grids co-resident: the library
                # reports the timed-out launch once and serves the device with the launch chain; repeat, as it says)
                try:
                    return fn0(i)
                except RuntimeError as e:
                    if "timed out inside the persistent kernel" not in str(e):
                        raise
                    return fn0(i)

            for i in range(min(args.warmup, 10)):
                fn(i)
            sync()
            t0 = time.perf_counter()
            for i in range(args.steps):
                fn(i)
            sync()
            tt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return 1e3 * float(tt.item()) / args.steps

        n_gt1 = {"overlap_mode_of_value": "overlapped" if overlap else "sequential",
                 "sequential_ms_per_step": timed(lambda i: op @ vs[i % nbuf]),
                 "shard_product_alone_ms": timed(lambda i: G @ vs[i % nbuf]),
                 "last_async_route": getattr(op, "async_route", None),
                 "note": "overlapped steps take the persistent kernel whenever the previous collective has completed "
                         "(event query), else the launch chain; sequential steps always take the persistent kernel"}
    result = {
        "metric": "curvature matvecs/s (GGN, D=10M MLP)",
        "value": world * args.steps / elapsed,
        "unit": "matvecs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        # weak scaling: one step = ONE product of the GGN of the whole (N x rows_per_gpu)-row data set with a vector,
        # computed as N shard products + one all-reduce.  `value` counts the shard products all ranks finished (the
        # contract's "units all ranks processed"); the number of full-data products per second is beside it.
        "full_data_matvecs_per_s": args.steps / elapsed,
        "shard_products_per_s": world * args.steps / elapsed,
        "value_definition": "shard products/s = n_gpus x full-data matvecs/s (weak scaling: rows_per_gpu fixed, the "
                            "data set grows with n_gpus); at n_gpus = 1 both are BASELINE's matvecs/s",
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "C2: GGNLinearOperator @ v, MLP 1024-2688-2688-10 (D=10010122), MSE mean, "
                        f"{args.batch} rows per GPU, K=1",
            "rows_per_gpu": args.batch,
            "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default; not set by the bench)"),
            "parallelism": (f"dp{world} (data shards + RCCL all-reduce of the [D] result"
                            + (", overlapped with the next product)" if overlap else ")")) if world > 1 else "single GPU",
        },
    }

    if n_gt1 is not None:
        result["n_gt1"] = n_gt1

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- roofline leg: same steps with per-kernel HIP events on the launch stream
        _hip.prof_enable(True)
        nprof = min(args.steps, 200)
        for i in range(nprof):
            step(i)
        torch.cuda.synchronize()
        prof = _hip.prof_collect()
        _hip.prof_enable(False)
        fam_kernels = {"persistent": ["mlp_mega_kernel"],
                       "fwd_mfma": ["fwd_mfma_first_kernel", "fwd_mfma_kernel"], "bwd_dprev": ["bwd_fused_kernel"],
                       "outer_all": ["outer_all_kernel"], "finish_head_fwd": ["head_fwd_kernel"],
                       "loss_head_bwd": ["head_bwd_kernel"]}
        kernels = {}
        for fam, r in prof.items():
            names = fam_kernels.get(fam, [fam])
            kernels[fam] = {
                "kernels": names, "launches_per_matvec": r["launches"] / nprof,
                "alg_bytes_per_launch": r["alg_bytes"] / max(r["launches"], 1),
                "avg_launch_us_hip_events": 1e3 * r["ms"] / max(r["launches"], 1),
                "rocprof_avg_kernel_us": rocprof_avg_us(names),
                "achieved_GBps": (r["alg_bytes"] / (r["ms"] * 1e-3) / 1e9) if r["ms"] > 0 else 0.0,
                "us_per_matvec": 1e3 * r["ms"] / nprof,
            }
        dom = max(kernels, key=lambda k: kernels[k]["us_per_matvec"])
        tr = [pmc_traffic_per_launch(k) for fam in kernels for k in fam_kernels.get(fam, [fam])]
        traffic = sum(t for t in tr if t) if any(tr) else None  # every kernel of the chain runs once per matvec
        achieved = 12 * D / (ms_per_step * 1e-3) / 1e9
        result["roofline"] = {
            "bound": "hbm",
            "kernel": "whole GGN matvec = " + str(int(round(sum(k["launches_per_matvec"] for k in kernels.values()))))
                      + " launch(es) per product (" + ", ".join(n for k in kernels.values() for n in k["kernels"]) + ")",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "alg_bytes_per_matvec": 12 * D,
            "definition": "12 D algorithmic bytes (theta, v, result once; SURVEY 8d) / step time of the timed region",
            "traffic": traffic,
            "traffic_source": (str(pmc_traffic_meta()[0]) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                               "passes of this command, FETCH doubled per the gfx950 note), summed over the kernels of one "
                               "matvec; NOT measured in this run") if traffic else None,
            "traffic_stale": pmc_traffic_meta()[1] if traffic else None,
            "dominant_kernel": dom,
            "dominant_kernel_frac": kernels[dom]["achieved_GBps"] / HBM_PEAK_GBPS,
            "kernels": kernels,
            "kernels_note": "avg_launch_us_hip_events = HIP-event interval around each launch on the launch stream "
                            "(includes ~2.5 us dispatch latency that rocprofv3 durations exclude); "
                            "rocprof_avg_kernel_us from the newest profiles/rNN_c2_n8_bench_kernel_stats.txt (same command)",
            "kernel_us_per_matvec_hip_events": sum(k["us_per_matvec"] for k in kernels.values()),
        }
        try:  # the headline line must be printed whatever happens in the untimed legs
            result["cpu_baseline"] = cpu_baseline(args.batch)
        except Exception as e:  # noqa: BLE001
            result["cpu_baseline"] = {"error": repr(e)}
        try:
            result["other_points"] = other_points(model, params, device, D)
        except Exception as e:  # noqa: BLE001
            result["other_points"] = {"error": repr(e)}
        kc = result["other_points"].get("k32_columns_rows8") if isinstance(result.get("other_points"), dict) else None
        if kc:
            result["roofline_kcols"] = {
                "bound": "hbm", "kernel": "K = 32 probe columns (clo_mlp_ggn_matmat: kfwd_stream / kouter_stream + GEMM chain)",
                "achieved": kc["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": kc["frac_of_hbm_peak"],
                "alg_bytes_per_column": 8 * D,
                "definition": "8 D algorithmic bytes per column (V read, result written; W shared; SURVEY 8d) x 32 columns / "
                              "time of one G @ [D, 32] product; K = 64 beside it in other_points",
                "profile": ", ".join("profiles/" + os.path.basename(latest_profile(f) or "?") for f in
                                     ("c2_k32_kernel_stats.txt", "c2_k32_pmc_traffic.txt")),
            }
        try:
            result["other_points"].update(secondary_configs(device))
        except Exception as e:  # noqa: BLE001
            result["other_points"]["secondary_configs_error"] = repr(e)

    if not args.no_extras:
        # ---- second half of the metric: KFAC factor build ms/batch (every rank takes part)
        try:
            result["kfac"] = kfac_leg(device, world, rank)
        except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
            result["kfac"] = {"error": repr(e)}
        if world > 1:
            # ---- the same operator where data-parallel matvecs can scale: 512 rows per rank
            try:
                rows = 512
                m2, X2, y2 = build_problem(device, rows, seed=100 + rank)
                p2 = dict(m2.named_parameters())
                G2 = AllReducedLinearOperator(C.GGNLinearOperator(m2, nn.MSELoss(), p2, [(X2, y2)],
                                                                  check_deterministic=False, num_data=rows * world))
                vv = [torch.rand(D, device=device) for _ in range(4)]
                pend: list = []

                def step2(i):
                    yv, work = G2.matmul_async(vv[i % 4])
                    pend.append(work)
                    if len(pend) > 2:
                        pend.pop(0).wait()

                for i in range(5):
                    step2(i)
                while pend:
                    pend.pop(0).wait()
                sync()
                t0 = time.perf_counter()
                nst = 40
                for i in range(nst):
                    step2(i)
                while pend:
                    pend.pop(0).wait()
                sync()
                el = time.perf_counter() - t0
                t = torch.tensor([el], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
                result["scalable_points"] = {"rows512": {
                    "shard_matvecs_per_s": world * nst / el, "ms_per_step": 1e3 * el / nst, "rows_per_gpu": rows,
                    "alg_tflops_total": world * 10.0 * rows * D * nst / el / 1e12,
                    "note": "same operator, 512 rows per rank (MFMA-bound), all-reduce of the [D] result overlapped "
                            "with the next product"}}
            except Exception as e:  # noqa: BLE001
                result["scalable_points"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
