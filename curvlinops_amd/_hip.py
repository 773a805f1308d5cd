"""ctypes binding of the C-ABI library ``libclo_hip.so`` (see ``include/curvlinops_amd.h``).

PyTorch only provides device memory and the HIP stream; every function here hands raw
device pointers to the hand-written gfx950 kernels.  There is deliberately NO fallback: if
the shared object is missing or a call fails, a ``RuntimeError`` is raised.
"""

from __future__ import annotations

from contextlib import contextmanager
import ctypes
import threading
from ctypes import POINTER, c_char_p, c_float, c_int, c_long, c_uint64, c_void_p
from pathlib import Path

import torch
from torch import Tensor

import os

# CLO_HIP_LIB lets kernel-ablation builds be A/B-tested; the default is the in-tree library.
_LIB_PATH = Path(os.environ.get("CLO_HIP_LIB") or Path(__file__).resolve().parent / "lib" / "libclo_hip.so")
_lib = None

ACT_IDENTITY, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
LOSS_MSE, LOSS_CE, LOSS_BCE, LOSS_RANK1 = 0, 1, 2, 3
LOSS_EF_MSE, LOSS_EF_CE, LOSS_EF_BCE = 4, 5, 6   # empirical Fisher, per-sample loss gradient formed in the kernel from the targets

# name -> (restype, argtypes); mirrors include/curvlinops_amd.h one to one
_PF = c_void_p  # float* passed as integer address
_SIGNATURES = {
    "clo_version": (c_int, []),
    "clo_last_error": (c_char_p, []),
    "clo_persistent_status": (c_int, [c_int]),
    "clo_fault_pending": (c_int, [c_int]),
    "clo_test_set_spin_limit": (c_int, [ctypes.c_uint]),
    "clo_test_occupy": (c_int, [c_int, c_int, c_long, c_void_p]),
    "clo_prof_enable": (c_int, [c_int]),
    "clo_prof_collect": (c_int, [POINTER(ctypes.c_double), POINTER(c_long), POINTER(ctypes.c_double)]),
    "clo_gemm_f32": (
        c_int,
        [c_int, c_int, c_int, c_float, _PF, c_long, c_long, c_long, _PF, c_long, c_long, c_long,
         c_float, _PF, c_long, c_long, c_int, c_int, _PF, c_void_p],
    ),
    "clo_gemm_suggest_splitk": (c_int, [c_int, c_int, c_int, c_int]),
    "clo_gemm_streamk_ws_floats": (c_long, []),
    "clo_syrk_suggest_splitk": (c_int, [c_int, c_long]),
    "clo_gemm_sqsum_f32": (
        c_int,
        [c_int, c_int, c_int, c_float, _PF, c_long, c_long, c_long, _PF, c_long, c_long, c_long,
         c_float, _PF, c_long, c_int, c_int, _PF, c_void_p],
    ),
    "clo_gemm_sqsum_suggest_splits": (c_int, [c_int, c_int, c_int]),
    "clo_potrf_diag_f32": (c_int, [_PF, c_long, c_int, _PF, c_long, c_void_p, c_int, c_void_p]),
    "clo_cholesky_inverse_f32": (c_int, [_PF, c_long, _PF, c_long, c_int, c_float, _PF, c_void_p, c_void_p]),
    "clo_cholesky_inverse_ws_floats": (c_long, [c_int]),
    "clo_cholesky_inverse_batched_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                                 _PF, c_void_p, c_void_p]),
    "clo_cholesky_inverse_batched_ws_floats": (c_long, [c_int, c_int]),
    "clo_sytrd_f32": (c_int, [_PF, c_long, c_int, _PF, _PF, _PF, _PF, c_long, c_int, c_void_p]),
    "clo_eigh_ws_bytes": (c_long, [c_int, c_int]),
    "clo_eigh_f32": (c_int, [_PF, c_long, c_int, _PF, _PF, c_long, c_void_p, c_long, c_int, c_void_p]),
    "clo_eigh_batched_f32": (c_int, [_PF, c_long, c_long, c_int, c_int, _PF, c_long, _PF, c_long, c_long, c_void_p, c_long,
                                     c_int, c_void_p]),
    "clo_stedc_ws_bytes": (c_long, [c_int, c_int]),
    "clo_stedc_f32": (c_int, [_PF, _PF, c_long, c_int, c_int, _PF, c_long, _PF, c_long, c_long, c_void_p, c_long, c_void_p]),
    "clo_sytrd_ws_bytes": (c_long, [c_int]),
    "clo_im2col_syrk_accum_f32": (
        c_int,
        [_PF, c_long, _PF, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_int, c_int, c_int, c_float, c_float, c_int, _PF, c_void_p],
    ),
    "clo_patch_fold_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "clo_patch_fold_f32": (
        c_int,
        [_PF, c_long, _PF, c_long, _PF, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_int, c_int, c_int, c_float, c_float, c_void_p],
    ),
    "clo_im2col_f32": (
        c_int,
        [_PF, _PF, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_int, c_int, c_void_p],
    ),
    "clo_gemm_ptrs_f32": (
        c_int,
        [c_int, c_int, c_int, c_float, _PF, POINTER(c_void_p), c_long, c_long, c_long, _PF, POINTER(c_void_p), c_long, c_long,
         c_long, c_float, _PF, c_long, c_long, c_int, c_int, _PF, c_void_p],
    ),
    "clo_syrk_accum_f32": (
        c_int,
        [_PF, c_long, _PF, c_long, c_int, c_long, c_int, c_float, c_float, c_int, _PF, c_void_p],
    ),
    "clo_mlp_fwd_jvp_layer": (
        c_int,
        [_PF, _PF, _PF, _PF, _PF, _PF, _PF, _PF, _PF, c_int, c_int, c_int, c_int, _PF, c_void_p],
    ),
    "clo_mlp_fwd_ws_floats": (c_long, [c_int, c_int, c_int]),
    "clo_loss_hessian_apply": (
        c_int,
        [c_int, _PF, _PF, c_int, _PF, _PF, _PF, c_int, c_int, c_float, c_void_p],
    ),
    "clo_mlp_bwd_layer": (
        c_int,
        [_PF, _PF, _PF, _PF, _PF, _PF, _PF, c_float, c_float, c_int, c_int, c_int, _PF, c_void_p],
    ),
    "clo_mlp_bwd_ws_floats": (c_long, [c_int, c_int, c_int]),
    "clo_mlp_ggn_matvec": (
        c_int,
        [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p),
         POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), _PF, c_int,
         c_int, _PF, c_int, c_float, c_float, c_float, c_int, _PF, c_void_p],
    ),
    "clo_mlp_ggn_ws_floats": (c_long, [c_int, POINTER(c_int), c_int]),
    "clo_mlp_ggn_ws_init": (c_int, [c_int, POINTER(c_int), c_int, _PF, c_void_p]),
    "clo_mlp_ggn_matmat": (
        c_int,
        [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p),
         POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_long, _PF, c_int,
         c_int, c_int, _PF, c_int, c_float, c_float, c_float, _PF, c_void_p],
    ),
    "clo_mlp_ggn_matmat_ws_floats": (c_long, [c_int, POINTER(c_int), c_int, c_int]),
    "clo_mlp_hessian_matmat": (
        c_int,
        [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p),
         POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_long, _PF, c_int,
         c_int, _PF, c_int, c_float, c_float, c_float, _PF, c_void_p],
    ),
    "clo_mlp_hessian_matmat_ws_floats": (c_long, [c_int, POINTER(c_int), c_int, c_int]),
    "clo_mlp_hessian_matvec": (
        c_int,
        [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p),
         POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), _PF, c_int, _PF,
         c_int, _PF, c_int, c_float, c_float, c_float, _PF, c_void_p],
    ),
    "clo_mlp_hessian_ws_floats": (c_long, [c_int, POINTER(c_int), c_int]),
    "clo_mlp_jac_ws_floats": (c_long, [c_int, POINTER(c_int), c_int]),
    "clo_mlp_jvp": (
        c_int,
        [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
         POINTER(c_void_p), _PF, c_int, _PF, _PF, c_void_p],
    ),
    "clo_mlp_loss_grad": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p), _PF, c_int, c_int,
                                  _PF, c_float, _PF, _PF, c_void_p]),
    "clo_mlp_vjp": (
        c_int,
        [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
         POINTER(c_void_p), _PF, c_int, _PF, c_float, c_float, _PF, c_void_p],
    ),
    "clo_gram_tall_supported": (c_int, [c_long, c_int, c_int]),
    "clo_gram_tall_ws_floats": (c_long, [c_long, c_int, c_int]),
    "clo_gram_tall_f32": (c_int, [_PF, c_long, _PF, c_long, c_int, c_long, c_int, c_float, c_float, _PF, c_void_p]),
    "clo_syrk_grouped_max_problems": (c_int, []),
    "clo_syrk_grouped_ws": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "clo_syrk_grouped_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, _PF, c_void_p, c_void_p]),
    "clo_tall_gram_ws_bytes": (c_long, [c_long, c_int, c_int]),
    "clo_tall_gram_f64": (c_int, [c_void_p, c_long, _PF, c_long, c_int, _PF, c_long, c_int, c_long, c_void_p, c_void_p]),
    "clo_tall_apply_f32": (c_int, [_PF, c_long, _PF, c_long, c_float, _PF, c_long, _PF, c_long, c_long, c_int, c_int,
                                   c_void_p]),
    "clo_axpby_f32": (c_int, [_PF, _PF, c_long, c_float, c_float, c_void_p]),
    "clo_dot_ws_bytes": (c_long, []),
    "clo_dot_f32": (c_int, [_PF, _PF, c_long, c_float, _PF, c_void_p, c_void_p]),
    "clo_cg_update_f32": (c_int, [_PF, _PF, _PF, _PF, c_long, _PF, _PF, _PF, c_void_p, c_void_p]),
    "clo_cg_direction_f32": (c_int, [_PF, _PF, c_long, _PF, _PF, c_void_p]),
    "clo_transpose_f32": (c_int, [_PF, _PF, c_long, c_long, c_void_p]),
    "clo_rowscale_f32": (c_int, [_PF, _PF, _PF, c_long, c_long, c_int, c_float, c_void_p]),
    "clo_canonical_pack_f32": (c_int, [_PF, _PF, _PF, c_long, c_long, c_long, c_void_p]),
    "clo_canonical_unpack_f32": (c_int, [_PF, _PF, _PF, c_long, c_long, c_long, c_void_p]),
    "clo_pack_probes_f32": (c_int, [_PF, c_long, c_long, c_uint64, c_int, c_void_p]),
    "clo_larft_f32": (c_int, [_PF, _PF, _PF, c_int, c_int, c_void_p]),
    "clo_ormtr_ws_floats": (c_long, [c_int, c_int]),
    "clo_ormtr_f32": (c_int, [_PF, c_long, _PF, _PF, c_long, c_int, c_int, _PF, c_long, c_void_p]),
    "clo_tql2_batched_f32": (c_int, [_PF, _PF, _PF, _PF, c_int, c_int, _PF, c_void_p]),
    "clo_dc_deflate": (c_int, [_PF, _PF, _PF, _PF, _PF, _PF, _PF, _PF, c_int, c_int, ctypes.c_double, c_void_p]),
    "clo_dc_secular": (c_int, [_PF, _PF, _PF, _PF, _PF, _PF, _PF, c_int, c_int, c_int, c_void_p]),
    "clo_dc_build": (c_int, [_PF, _PF, _PF, _PF, _PF, _PF, _PF, c_int, c_int, c_int, c_void_p]),
    "clo_dc_rotate": (c_int, [_PF, _PF, _PF, _PF, c_int, c_int, c_void_p]),
    "clo_kron_ws_floats": (c_long, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "clo_ekfac_correction_ws_floats": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "clo_ekfac_correction_f32": (c_int, [_PF, c_long, _PF, c_long, _PF, c_long, c_int, _PF, _PF, c_int, c_int, c_int, c_int,
                                         c_int, c_float, c_float, _PF, c_long, c_void_p]),
    "clo_kron_matmat": (c_int, [_PF, _PF, c_long, _PF, c_long, _PF, c_int, c_int, c_int, c_int, c_int, c_int, _PF, c_long, c_void_p]),
    "clo_eigh_apply": (c_int, [_PF, _PF, c_long, _PF, c_long, _PF, _PF, c_int, c_int, c_int, c_int, _PF, c_long, c_void_p]),
    "clo_kron_matmat_blocks": (
        c_int,
        [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_long), POINTER(c_void_p), POINTER(c_long), POINTER(c_void_p),
         POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_int, _PF, c_long,
         c_void_p],
    ),
}


def exported_symbols() -> list[str]:
    """Names every build of the library must export (checked by the CPU test-suite)."""
    return list(_SIGNATURES)


def lib_path() -> Path:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load the shared object once; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: run `python -m curvlinops_amd.csrc.build` "
                "(hipcc, gfx950). curvlinops_amd has no fallback path."
            )
        lib = ctypes.CDLL(str(_LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            if os.environ.get("CLO_HIP_LIB") and not hasattr(lib, name):
                continue   # (A/B runs against an older build of the library: entry points added since are absent)
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            if res is c_int and args and args[-1] is c_void_p:   # status-returning entry points that take a stream
                setattr(lib, name, _device_safe(fn))
        _lib = lib
    return _lib


def _device_safe(fn):
    """`_stream()` may have made another device current for the duration of the foreign call and `_check()`
    switches back; if the call itself raises (ctypes argument conversion), nothing would -- restore here."""

    def call(*args):
        try:
            return fn(*args)
        except BaseException:
            prev = getattr(_tls, "restore", None)
            if prev is not None:
                _tls.restore = None
                torch.cuda.set_device(prev)
            raise

    call.__name__ = getattr(fn, "__name__", "clo_call")
    return call


def has(symbol: str) -> bool:
    """Whether this build of the library exports ``symbol``."""
    return symbol in _SIGNATURES and hasattr(load(), symbol)


def persistent_status(device: int | None = None) -> int:
    """Bit mask of the co-resident-grid modes the library has DISABLED on a device after one of their launches timed out
    (bit 0 persistent MLP kernel, bit 1 tridiagonalisation panels, bit 2 stream-K GEMM): ``clo_persistent_status``."""
    dev = torch.cuda.current_device() if device is None else int(device)
    return int(load().clo_persistent_status(dev))


def raise_if_async_fault(device: int | None = None) -> None:
    """For consumers that have just synchronised with the device (host copies of a result): a persistent / stream-K launch
    that timed out since the last report makes THIS result suspect -- raise now instead of on the next call
    (``clo_fault_pending``; the timed-out launch has also marked its own output with NaN)."""
    if not torch.cuda.is_available():
        return
    dev = torch.cuda.current_device() if device is None else int(device)
    bits = int(load().clo_fault_pending(dev))
    if bits:
        names = [n for k, n in enumerate(("persistent MLP kernel", "tridiagonalisation panels", "stream-K GEMM")) if bits >> k & 1]
        raise RuntimeError(f"curvlinops_amd: a launch on device {dev} timed out waiting for its own workgroups "
                           f"({', '.join(names)}; GPU shared with another process or CU-masked).  The result that was "
                           "just read may be invalid (it carries NaN marks); repeat the call -- the library falls back "
                           "to the multi-launch route on this device.")


def prof_enable(on: bool) -> None:
    _check(load().clo_prof_enable(int(on)), "clo_prof_enable")


def prof_collect() -> dict[str, dict[str, float]]:
    """Per-kernel-family totals recorded since the last call: ms, launches, algorithmic bytes."""
    ms = (ctypes.c_double * 8)()
    cnt = (c_long * 8)()
    by = (ctypes.c_double * 8)()
    _check(load().clo_prof_collect(ms, cnt, by), "clo_prof_collect")
    # tags of csrc/clo_common.h:ProfScope as used by mlp.hip
    names = ["fwd_mfma", "loss_head_bwd", "bwd_dprev", "finish_head_fwd", "outer_all", "other", "persistent", "t7"]
    return {n: {"ms": ms[i], "launches": int(cnt[i]), "alg_bytes": by[i]} for i, n in enumerate(names) if cnt[i]}


_tls = threading.local()


def _check(rc: int, what: str) -> None:
    prev = getattr(_tls, "restore", None)
    if prev is not None:  # `_stream()` switched the current device for this call: switch back
        _tls.restore = None
        torch.cuda.set_device(prev)
    if rc != 0:
        msg = load().clo_last_error().decode(errors="replace")
        kind = {-1: ValueError, -3: RuntimeError}.get(rc, RuntimeError)
        raise kind(f"{what} failed (code {rc}): {msg}")


def _stream() -> int:
    """The current HIP stream of the device the call's operands live on (the device of the last
    tensor `_p` converted in this thread; argument lists convert their tensors before the stream).
    If that is not the current device -- an operator on ``cuda:1`` used while ``cuda:0`` is current --
    the device is made current for the duration of the foreign call (`_check` switches back): kernels
    must be launched with their stream's device current."""
    dev = getattr(_tls, "dev", None)
    _tls.dev = None
    if dev is None:
        return torch.cuda.current_stream().cuda_stream
    cur = torch.cuda.current_device()
    if dev.index is not None and dev.index != cur:
        _tls.restore = cur
        torch.cuda.set_device(dev)
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t: Tensor | None) -> int | None:
    """Device address of a float32 CUDA(HIP) tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32:
        raise ValueError(f"expected a float32 GPU tensor, got {t.dtype} on {t.device}")
    _tls.dev = t.device
    return t.data_ptr()


def _pc(t: Tensor | None) -> int | None:
    if t is not None and not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    return _p(t)


# --------------------------------------------------------------------------------------
# GEMM family
# --------------------------------------------------------------------------------------
def gemm(A: Tensor, B: Tensor, out: Tensor | None = None, alpha: float = 1.0, beta: float = 0.0,
         splitk: int | None = None) -> Tensor:
    """``out = alpha * A @ B + beta * out`` for 2-D or batched 3-D fp32 GPU tensors.

    ``A``/``B`` may be arbitrary strided views (e.g. ``.T`` / ``.mT``); ``out`` must be
    row-major (last stride 1).
    """
    lib = load()
    batched = A.dim() == 3 or B.dim() == 3
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    B3 = B if B.dim() == 3 else B.unsqueeze(0)
    nb = max(A3.shape[0], B3.shape[0])
    M, K = A3.shape[1], A3.shape[2]
    K2, N = B3.shape[1], B3.shape[2]
    if K != K2 or A3.shape[0] not in (1, nb) or B3.shape[0] not in (1, nb):
        raise ValueError(f"gemm shape mismatch: {tuple(A.shape)} @ {tuple(B.shape)}")
    if out is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs an `out` tensor")
        out = torch.empty((nb, M, N) if batched else (M, N), device=A.device, dtype=torch.float32)
    O3 = out if out.dim() == 3 else out.unsqueeze(0)
    if O3.shape != (nb, M, N) or (N > 1 and O3.stride(2) != 1) or O3.stride(1) < N:
        if not (N == 1 and O3.shape == (nb, M, N)):
            raise ValueError(f"bad out tensor {tuple(out.shape)} / strides {out.stride()}")
    if M == 0 or N == 0:
        return out
    if K == 0:
        if beta == 0.0:
            out.zero_()
        else:
            out.mul_(beta)
        return out
    sa_b = A3.stride(0) if A3.shape[0] == nb and nb > 1 else 0
    sb_b = B3.stride(0) if B3.shape[0] == nb and nb > 1 else 0
    if splitk is None:
        splitk = lib.clo_gemm_suggest_splitk(M, N, K, nb)
    ws = None
    if splitk > 1:
        ws = torch.empty(nb * splitk * M * N, device=A.device, dtype=torch.float32)
    elif splitk < 0:  # stream-K schedule: partial tiles are finished inside the kernel
        ws = torch.empty(lib.clo_gemm_streamk_ws_floats(), device=A.device, dtype=torch.float32)
    ldc = O3.stride(1) if M > 1 else max(N, O3.stride(1))
    rc = lib.clo_gemm_f32(
        M, N, K, alpha, _p(A3), A3.stride(1), A3.stride(2), sa_b, _p(B3), B3.stride(1),
        B3.stride(2), sb_b, beta, _p(O3), ldc, O3.stride(0) if nb > 1 else 0, nb, splitk, _p(ws),
        _stream(),
    )
    _check(rc, "clo_gemm_f32")
    return out


GEMM_PTRS_MAX = 8   # members of one clo_gemm_ptrs_f32 launch


def gemm_members(A, B, out: Tensor | None = None, alpha: float = 1.0, beta: float = 0.0) -> Tensor:
    """Batched ``out[i] = alpha * A[i] @ B[i] + beta * out[i]`` where ``A`` and / or ``B`` is a LIST of equally shaped and
    equally strided 2-D fp32 GPU tensors lying anywhere in memory (``clo_gemm_ptrs_f32``: no stacked copies), the other a
    3-D tensor (or also a list).  ``out``: 3-D, row-major matrices."""
    lib = load()
    la, lb = isinstance(A, (list, tuple)), isinstance(B, (list, tuple))
    n = len(A) if la else (len(B) if lb else A.shape[0])
    a0, b0 = (A[0] if la else A[0]), (B[0] if lb else B[0])
    M, K = a0.shape
    K2, N = b0.shape
    if K != K2:
        raise ValueError(f"gemm_members shape mismatch: {tuple(a0.shape)} @ {tuple(b0.shape)}")
    for lst in ([A] if la else []) + ([B] if lb else []):
        if len(lst) != n or any(t.shape != lst[0].shape or t.stride() != lst[0].stride() for t in lst):
            raise ValueError("gemm_members: list members must agree in shape and strides")
    if out is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs an `out` tensor")
        out = torch.empty(n, M, N, device=a0.device, dtype=torch.float32)
    if out.shape != (n, M, N) or (N > 1 and out.stride(2) != 1):
        raise ValueError(f"bad out tensor {tuple(out.shape)} / strides {out.stride()}")
    if M == 0 or N == 0 or n == 0:
        return out
    for i0 in range(0, n, GEMM_PTRS_MAX):
        cnt = min(GEMM_PTRS_MAX, n - i0)
        ap = (c_void_p * cnt)(*[t.data_ptr() for t in A[i0:i0 + cnt]]) if la else None
        bp = (c_void_p * cnt)(*[t.data_ptr() for t in B[i0:i0 + cnt]]) if lb else None
        a_base, b_base = (None if la else A[i0]), (None if lb else B[i0])
        splitk = lib.clo_gemm_suggest_splitk(M, N, K, cnt)
        ws = None
        if splitk > 1:
            ws = torch.empty(cnt * splitk * M * N, device=a0.device, dtype=torch.float32)
        elif splitk < 0:
            ws = torch.empty(lib.clo_gemm_streamk_ws_floats(), device=a0.device, dtype=torch.float32)
        o = out[i0:i0 + cnt]
        _p(o)   # (sets the device the launch belongs to)
        rc = lib.clo_gemm_ptrs_f32(
            M, N, K, alpha, None if la else a_base.data_ptr(), ap, a0.stride(0), a0.stride(1), 0 if la else A.stride(0),
            None if lb else b_base.data_ptr(), bp, b0.stride(0), b0.stride(1), 0 if lb else B.stride(0), beta,
            o.data_ptr(), o.stride(1) if M > 1 else max(N, o.stride(1)), o.stride(0) if cnt > 1 else 0, cnt, splitk, _p(ws),
            _stream())
        _check(rc, "clo_gemm_ptrs_f32")
    return out


def gemm_sqsum(A: Tensor, B: Tensor, out: Tensor, alpha: float = 1.0, beta: float = 0.0) -> Tensor:
    """``out = beta*out + alpha * sum_b (A[b] @ B[b])**2`` for batched ``A [nb,M,K]``, ``B [nb,K,N]``
    (strided views allowed), ``out [M,N]`` row-major."""
    lib = load()
    nb, M, K = A.shape
    nb2, K2, N = B.shape
    if nb != nb2 or K != K2 or out.shape != (M, N) or (N > 1 and out.stride(1) != 1):
        raise ValueError(f"gemm_sqsum shape mismatch: {tuple(A.shape)} {tuple(B.shape)} -> {tuple(out.shape)}")
    splits = lib.clo_gemm_sqsum_suggest_splits(M, N, nb)
    ws = torch.empty(splits * M * N, device=A.device, dtype=torch.float32) if splits > 1 else None
    rc = lib.clo_gemm_sqsum_f32(M, N, K, alpha, _p(A), A.stride(1), A.stride(2), A.stride(0), _p(B),
                                B.stride(1), B.stride(2), B.stride(0), beta, _p(out),
                                out.stride(0) if M > 1 else max(N, out.stride(0)), nb, splits, _p(ws),
                                _stream())
    _check(rc, "clo_gemm_sqsum_f32")
    return out


def syrk_accum(C: Tensor, X: Tensor, alpha: float = 1.0, beta: float = 1.0, ones_col: bool = False,
               splitk: int | None = None, force_gram_tall: bool = False) -> Tensor:
    """``C = beta*C + alpha * [X|1]^T [X|1]`` for row-major ``X[rows, d]`` (view with stride ok).
    ``force_gram_tall``: run the streaming tall-skinny kernel whatever the dispatch policy says (tests)."""
    lib = load()
    if X.dim() != 2 or (X.shape[1] > 1 and X.stride(1) != 1):
        raise ValueError("X must be 2-D with unit column stride")
    rows, d = X.shape
    dd = d + (1 if ones_col else 0)
    if C.shape != (dd, dd) or C.stride(1) != 1:
        raise ValueError(f"C must be [{dd},{dd}] row-major, got {tuple(C.shape)}")
    ldx = X.stride(0) if rows > 1 else max(d, 1)
    if force_gram_tall and not (1 <= dd <= 128):
        raise ValueError("the tall-skinny Gram kernel needs 1 <= d + ones_col <= 128")
    if force_gram_tall or (splitk is None and lib.clo_gram_tall_supported(rows, d, int(ones_col))):
        # tall and skinny (conv-layer factors, Hutch++ Gram passes): the streaming Gram kernel
        ws = torch.empty(lib.clo_gram_tall_ws_floats(rows, d, int(ones_col)), device=X.device, dtype=torch.float32)
        rc = lib.clo_gram_tall_f32(_p(C), C.stride(0), _p(X), rows, d, ldx, int(ones_col), alpha, beta, _p(ws),
                                   _stream())
        _check(rc, "clo_gram_tall_f32")
        return C
    if splitk is None:
        splitk = lib.clo_syrk_suggest_splitk(dd, rows)
    ws = torch.empty(splitk * dd * dd, device=X.device, dtype=torch.float32) if splitk > 1 else None
    rc = lib.clo_syrk_accum_f32(_p(C), C.stride(0), _p(X), rows, d, ldx, int(ones_col), alpha, beta,
                                splitk, _p(ws), _stream())
    _check(rc, "clo_syrk_accum_f32")
    return C


def syrk_grouped(Cs: list[Tensor], Xs: list[Tensor], alphas: list[float], betas: list[float],
                 ones_cols: list[bool] | None = None) -> None:
    """``Cs[p] = betas[p] Cs[p] + alphas[p] [Xs[p] | 1]^T [Xs[p] | 1]`` for all p in ONE launch per group of up to
    ``clo_syrk_grouped_max_problems`` problems (``clo_syrk_grouped_f32``): the small covariance products of a KFAC factor
    build.  ``Xs[p]``: fp32 GPU ``[rows, d]`` with unit column stride; ``Cs[p]``: distinct ``[d(+1), d(+1)]`` row-major."""
    import ctypes

    lib = load()
    n = len(Cs)
    if n == 0:
        return
    ones_cols = [False] * n if ones_cols is None else list(ones_cols)
    cap = int(lib.clo_syrk_grouped_max_problems())
    for lo in range(0, n, cap):
        hi = min(n, lo + cap)
        P = hi - lo
        C, X = Cs[lo:hi], [x if (x.dim() == 2 and (x.shape[1] <= 1 or x.stride(1) == 1)) else x.contiguous() for x in Xs[lo:hi]]
        for c, x, o in zip(C, X, ones_cols[lo:hi]):
            dd = x.shape[1] + (1 if o else 0)
            if x.dim() != 2 or tuple(c.shape) != (dd, dd) or c.stride(1) != 1 or not (is_f32_gpu(c) and is_f32_gpu(x)):
                raise ValueError(f"syrk_grouped: X {tuple(x.shape)} does not fit C {tuple(c.shape)} (fp32 GPU, row-major)")
        ptr_t, long_t, int_t, float_t = ctypes.c_void_p * P, ctypes.c_long * P, ctypes.c_int * P, ctypes.c_float * P
        rows = long_t(*[x.shape[0] for x in X])
        d = int_t(*[x.shape[1] for x in X])
        ones = int_t(*[int(o) for o in ones_cols[lo:hi]])
        slab_n, cnt_n = ctypes.c_long(0), ctypes.c_long(0)
        _check(lib.clo_syrk_grouped_ws(P, rows, d, ones, ctypes.byref(slab_n), ctypes.byref(cnt_n)), "clo_syrk_grouped_ws")
        dev = X[0].device
        slab = torch.empty(max(1, slab_n.value), device=dev, dtype=torch.float32)
        cnt = torch.zeros(max(1, cnt_n.value), device=dev, dtype=torch.int32)
        _tls.dev = dev
        rc = lib.clo_syrk_grouped_f32(
            P, ptr_t(*[c.data_ptr() for c in C]), long_t(*[c.stride(0) for c in C]), ptr_t(*[x.data_ptr() for x in X]), rows, d,
            long_t(*[x.stride(0) if x.shape[0] > 1 else max(x.shape[1], 1) for x in X]), ones,
            float_t(*[float(a) for a in alphas[lo:hi]]), float_t(*[float(b) for b in betas[lo:hi]]), slab.data_ptr(),
            cnt.data_ptr(), _stream())
        _check(rc, "clo_syrk_grouped_f32")


def tall_gram_supported(X: Tensor, Y: Tensor | None = None) -> bool:
    ok = is_f32_gpu(X) and X.dim() == 2 and 1 <= X.shape[1] <= 64 and (X.shape[1] == 1 or X.stride(1) == 1)
    if Y is not None:
        ok = ok and is_f32_gpu(Y) and Y.dim() == 2 and Y.shape[0] == X.shape[0] and 1 <= Y.shape[1] <= 64 \
            and (Y.shape[1] == 1 or Y.stride(1) == 1) and Y.device == X.device
    return bool(ok)


def is_f32_gpu(t: Tensor) -> bool:
    return isinstance(t, Tensor) and t.is_cuda and t.dtype == torch.float32


def tall_gram(X: Tensor, Y: Tensor | None = None) -> Tensor:
    """``X^T Y`` (``X^T X`` without ``Y``) of tall float32 blocks ``[m, <= 64]`` as a FLOAT64 ``[n1, n2]`` matrix: one
    streaming pass, exact products, float64 accumulation (``clo_tall_gram_f64``)."""
    lib = load()
    m, n1 = X.shape
    n2 = n1 if Y is None else Y.shape[1]
    out = torch.empty(n1, n2, device=X.device, dtype=torch.float64)
    ws = torch.empty(max(1, lib.clo_tall_gram_ws_bytes(m, n1, n2) // 8), device=X.device, dtype=torch.float64)
    ldx = X.stride(0) if m > 1 else max(n1, 1)
    ldy = 0 if Y is None else (Y.stride(0) if m > 1 else max(n2, 1))
    px = _p(X)
    rc = lib.clo_tall_gram_f64(out.data_ptr(), n2, px, ldx, n1, None if Y is None else Y.data_ptr(), ldy, n2, m,
                               ws.data_ptr(), _stream())
    _check(rc, "clo_tall_gram_f64")
    return out


def tall_apply_supported(Q: Tensor, C: Tensor, G: Tensor | None = None) -> bool:
    ok = (is_f32_gpu(Q) and is_f32_gpu(C) and Q.dim() == 2 and C.dim() == 2 and Q.shape[1] == C.shape[0]
          and 1 <= Q.shape[1] <= 64 and Q.shape[1] % 4 == 0 and 1 <= C.shape[1] <= 64 and Q.stride(1) == 1
          and (Q.shape[0] <= 1 or Q.stride(0) % 4 == 0) and Q.data_ptr() % 16 == 0 and C.is_contiguous())
    if G is not None:
        ok = ok and is_f32_gpu(G) and G.shape == (Q.shape[0], C.shape[1]) and (G.shape[1] == 1 or G.stride(1) == 1)
    return bool(ok)


def tall_apply(Q: Tensor, C: Tensor, G: Tensor | None = None, beta: float = 1.0, out: Tensor | None = None) -> Tensor:
    """``beta G + Q C`` for a tall ``Q [m, k]`` and a small ``C [k, n]`` (k, n <= 64) in one pass over ``Q``, ``G`` and the
    result (``clo_tall_apply_f32``; rows = the M dimension of 16x16x4 MFMA tiles, ``C`` in registers)."""
    lib = load()
    m, k = Q.shape
    n = C.shape[1]
    if out is None:
        out = torch.empty(m, n, device=Q.device, dtype=torch.float32)
    ldq = Q.stride(0) if m > 1 else max(k, 4)
    rc = lib.clo_tall_apply_f32(_p(out), out.stride(0) if m > 1 else n, None if G is None else G.data_ptr(),
                                0 if G is None else (G.stride(0) if m > 1 else n), float(beta), _p(Q), ldq, _p(C), n, m, k,
                                n, _stream())
    _check(rc, "clo_tall_apply_f32")
    return out


def im2col(x: Tensor, kernel_size, stride, padding, dilation) -> Tensor:
    """Patches of a ``[B, C, H, W]`` fp32 GPU tensor as ``[B, OH*OW, C*KH*KW]`` (the layout of
    ``unfold(x).transpose(1, 2)``) in ONE launch for the whole mini-batch."""
    B, C_, H, W = x.shape
    (KH, KW), (SH, SW), (PH, PW), (DH, DW) = kernel_size, stride, padding, dilation
    OH = (H + 2 * PH - DH * (KH - 1) - 1) // SH + 1
    OW = (W + 2 * PW - DW * (KW - 1) - 1) // SW + 1
    out = torch.empty(B, OH * OW, C_ * KH * KW, device=x.device, dtype=torch.float32)
    rc = load().clo_im2col_f32(_pc(x.contiguous()), _pc(out), B, C_, H, W, KH, KW, SH, SW, PH, PW, DH, DW,
                               OH, OW, _stream())
    _check(rc, "clo_im2col_f32")
    return out


def pixel_gram_accum(C: Tensor, x: Tensor, kernel_size, stride, padding, dilation, alpha: float = 1.0,
                      beta: float = 1.0, ones_col: bool = False) -> Tensor:
    """``C = beta C + alpha [P | 1]^T [P | 1]`` with ``P = unfold(x)^T`` WITHOUT forming patches: the pixel Gram
    ``Gam = X^T X`` of ``X = x.view(B, C*H*W)`` (one dense SYRK with K = B) folded into the patch covariance by
    ``clo_patch_fold_f32`` (see ``csrc/conv.hip``).  For feature maps with ``(H W)^2 < OH OW (KH KW)^2``."""
    lib = load()
    B, C_, H, W = x.shape
    (KH, KW), (SH, SW), (PH, PW), (DH, DW) = kernel_size, stride, padding, dilation
    OH = (H + 2 * PH - DH * (KH - 1) - 1) // SH + 1
    OW = (W + 2 * PW - DW * (KW - 1) - 1) // SW + 1
    dd = C_ * KH * KW + (1 if ones_col else 0)
    if C.shape != (dd, dd) or C.stride(1) != 1:
        raise ValueError(f"C must be a row-major [{dd}, {dd}] matrix, got {tuple(C.shape)}")
    n = C_ * H * W
    ld = (n + 3) // 4 * 4
    X2 = x.contiguous().view(B, n)
    gam = torch.empty(n, ld, device=x.device, dtype=torch.float32)
    syrk_accum(gam[:, :n], X2, alpha=1.0, beta=0.0)
    colsum = X2.sum(dim=0) if ones_col else None
    rc = lib.clo_patch_fold_f32(_p(C), C.stride(0), _p(gam), ld, _pc(colsum), B, C_, H, W, KH, KW, SH, SW, PH, PW, DH, DW,
                                OH, OW, int(ones_col), alpha, beta, _stream())
    _check(rc, "clo_patch_fold_f32")
    return C


def im2col_syrk_accum(C: Tensor, x: Tensor, kernel_size, stride, padding, dilation, alpha: float = 1.0,
                      beta: float = 1.0, ones_col: bool = False) -> Tensor:
    """``C = beta C + alpha [P | 1]^T [P | 1]`` with ``P = unfold(x)^T`` (``[B*OH*OW, C*KH*KW]``) generated
    inside the GEMM's tile loader: the patch matrix is never materialised."""
    lib = load()
    B, C_, H, W = x.shape
    (KH, KW), (SH, SW), (PH, PW), (DH, DW) = kernel_size, stride, padding, dilation
    OH = (H + 2 * PH - DH * (KH - 1) - 1) // SH + 1
    OW = (W + 2 * PW - DW * (KW - 1) - 1) // SW + 1
    dd = C_ * KH * KW + (1 if ones_col else 0)
    if C.shape != (dd, dd) or C.stride(1) != 1:
        raise ValueError(f"C must be a row-major [{dd}, {dd}] matrix, got {tuple(C.shape)}")
    rows = B * OH * OW
    splitk = lib.clo_syrk_suggest_splitk(dd, rows) if rows > 0 else 1
    ws = torch.empty(splitk * dd * dd, device=x.device, dtype=torch.float32) if splitk > 1 else None
    rc = lib.clo_im2col_syrk_accum_f32(_p(C), C.stride(0), _pc(x.contiguous()), B, C_, H, W, KH, KW, SH, SW,
                                       PH, PW, DH, DW, OH, OW, int(ones_col), alpha, beta, splitk, _p(ws),
                                       _stream())
    _check(rc, "clo_im2col_syrk_accum_f32")
    return C


def cholesky_inverse_async(A: Tensor, damping: float = 0.0) -> tuple[Tensor, Tensor]:
    """Enqueue ``(A + damping I)^-1`` on the current stream: returns ``(out, status)`` where
    ``status`` is a device int32 (0 = ok, else the 1-based non-positive pivot) that the caller
    inspects once the stream has been joined.  One call into ``clo_cholesky_inverse_f32``
    (recursive blocked Cholesky carrying the triangular inverse; leaves in LDS, every O(n^3) step on
    the MFMA GEMM); never modifies ``A``."""
    n = A.shape[0]
    if A.dim() != 2 or A.shape[1] != n:
        raise ValueError(f"expected a square matrix, got {tuple(A.shape)}")
    out = torch.empty(n, n, device=A.device, dtype=torch.float32)
    status = torch.zeros(1, device=A.device, dtype=torch.int32)
    cholesky_inverse_into(A, damping, out, status)
    return out, status


def cholesky_inverse_into(A: Tensor, damping: float, out: Tensor, status: Tensor) -> None:
    """:func:`cholesky_inverse_async` into caller-provided ``out`` ([n, n] contiguous float32) and
    ``status`` (device int32), on the calling thread's current stream.  The whole chain of launches
    happens inside ONE foreign call (the GIL is released), so several host threads can drive
    independent factors concurrently."""
    lib = load()
    n = A.shape[0]
    if n == 0:
        return
    if A.stride(-1) != 1 and n > 1:
        A = A.contiguous()
    ws = torch.empty(lib.clo_cholesky_inverse_ws_floats(n), device=A.device, dtype=torch.float32)
    rc = lib.clo_cholesky_inverse_f32(_p(A), A.stride(0) if n > 1 else 1, _p(out), n, n, damping, _p(ws),
                                      status.data_ptr(), _stream())
    _check(rc, "clo_cholesky_inverse_f32")


def cholesky_inverse_batched_into(As: list[Tensor], dampings: list[float], outs: list[Tensor],
                                  status: Tensor) -> None:
    """``outs[b] = (As[b] + dampings[b] I)^-1`` for equally sized fp32 GPU factors in ONE chain of
    launches (``clo_cholesky_inverse_batched_f32``); ``status``: device int32 ``[len(As)]``."""
    import ctypes

    lib = load()
    n, nb = As[0].shape[0], len(As)
    if n == 0 or nb == 0:
        return
    As = [A if (A.stride(-1) == 1 or n == 1) else A.contiguous() for A in As]
    ptr_t, long_t, float_t = ctypes.c_void_p * nb, ctypes.c_long * nb, ctypes.c_float * nb
    a_ptrs = ptr_t(*[A.data_ptr() for A in As])
    ldas = long_t(*[A.stride(0) if n > 1 else 1 for A in As])
    o_ptrs = ptr_t(*[o.data_ptr() for o in outs])
    ldos = long_t(*[o.stride(0) if n > 1 else 1 for o in outs])
    damps = float_t(*[float(d) for d in dampings])
    ws = torch.empty(lib.clo_cholesky_inverse_batched_ws_floats(n, nb), device=As[0].device, dtype=torch.float32)
    rc = lib.clo_cholesky_inverse_batched_f32(a_ptrs, ldas, o_ptrs, ldos, n, nb, damps, _p(ws), status.data_ptr(),
                                              _stream())
    _check(rc, "clo_cholesky_inverse_batched_f32")


# cap on the workgroups of one persistent panel launch of clo_sytrd_f32 (0 = the library's own: one per compute unit of the
# device, at most 256): the eigensolver's multi-stream driver lowers it so that the launches running side by side stay co-resident
SYTRD_MAX_BLOCKS = 0


def sytrd_(A: Tensor, n: int, max_blocks: int | None = None) -> tuple[Tensor, Tensor, Tensor]:
    """In-place Householder tridiagonalisation of the symmetric matrix in ``A[:n, :n]`` (``clo_sytrd_f32``).
    ``A``: fp32 GPU tensor ``[>= n, ld]`` with ``ld % 4 == 0`` and zero padding columns.  Returns
    ``(D, E, tau)``; afterwards ``A`` holds the Householder vectors in LAPACK's ``uplo='L'`` layout of
    the column-major matrix."""
    lib = load()
    D = torch.empty(n, device=A.device, dtype=torch.float32)
    E = torch.empty(n, device=A.device, dtype=torch.float32)
    tau = torch.empty(n, device=A.device, dtype=torch.float32)
    nbytes = lib.clo_sytrd_ws_bytes(n)
    ws = torch.zeros(nbytes // 4, device=A.device, dtype=torch.float32)
    rc = lib.clo_sytrd_f32(_p(A), A.stride(0), n, _p(D), _p(E), _p(tau), _p(ws), nbytes,
                           SYTRD_MAX_BLOCKS if max_blocks is None else max_blocks, _stream())
    _check(rc, "clo_sytrd_f32")
    return D, E, tau


def not_pd_error(pivot: int, n: int) -> RuntimeError:
    return RuntimeError(f"cholesky: the input is not positive-definite (pivot {pivot} of {n} is not positive).")


def cholesky_inverse(A: Tensor, damping: float = 0.0) -> Tensor:
    """Synchronous form of :func:`cholesky_inverse_async`: raises ``RuntimeError`` if the matrix
    is not positive definite."""
    out, status = cholesky_inverse_async(A, damping)
    bad = int(status.item())
    if bad:
        raise not_pd_error(bad, A.shape[0])
    return out


# --------------------------------------------------------------------------------------
# MLP fast path
# --------------------------------------------------------------------------------------
def mlp_fwd_jvp_layer(W, b, VW, Vb, a_in, da_in, act: int):
    lib = load()
    N, d_in = a_in.shape
    d_out = W.shape[0]
    a_out = torch.empty(N, d_out, device=W.device, dtype=torch.float32)
    dphi = torch.empty_like(a_out)
    da_out = torch.empty_like(a_out) if (VW is not None or da_in is not None) else None
    ws = torch.empty(lib.clo_mlp_fwd_ws_floats(N, d_in, d_out), device=W.device, dtype=torch.float32)
    rc = lib.clo_mlp_fwd_jvp_layer(_pc(W), _pc(b), _pc(VW), _pc(Vb), _pc(a_in), _pc(da_in),
                                   _pc(a_out), _pc(da_out), _pc(dphi), N, d_in, d_out, act,
                                   _pc(ws), _stream())
    _check(rc, "clo_mlp_fwd_jvp_layer")
    return a_out, da_out, dphi


def loss_hessian_apply(kind: int, f, u, scale: float, aux=None, dphi_last=None):
    lib = load()
    N, C = f.shape
    w = torch.empty_like(f)
    rank = 1 if aux is None else (aux.shape[1] if aux.dim() == 3 else 1)
    rc = lib.clo_loss_hessian_apply(kind, _pc(f), _pc(aux), rank, _pc(u), _pc(dphi_last), _pc(w),
                                    N, C, scale, _stream())
    _check(rc, "clo_loss_hessian_apply")
    return w


def mlp_bwd_layer(W, delta, a_prev, dphi_prev, out_W, out_b, alpha: float, beta: float,
                  want_delta_prev: bool):
    lib = load()
    N, d_out = delta.shape
    d_in = W.shape[1]
    dprev = ws = None
    if want_delta_prev:
        dprev = torch.empty(N, d_in, device=W.device, dtype=torch.float32)
        ws = torch.empty(lib.clo_mlp_bwd_ws_floats(N, d_in, d_out), device=W.device,
                         dtype=torch.float32)
    rc = lib.clo_mlp_bwd_layer(_pc(W), _pc(delta), _pc(a_prev), _pc(dphi_prev), _pc(out_W),
                               _pc(out_b), _pc(dprev), alpha, beta, N, d_in, d_out, _pc(ws),
                               _stream())
    _check(rc, "clo_mlp_bwd_layer")
    return dprev


MLP_DEFAULT, MLP_NO_PERSISTENT = 0, 1  # CLO_MLP_* flags of clo_mlp_ggn_matvec


class MLPPlan:
    """Pre-marshalled argument tables for ``clo_mlp_ggn_matvec`` (one per operator).

    ``flags`` (``MLP_*``) is the kernel choice handed to every single-vector product of this plan: an
    argument of the C call, owned by the operator that owns the plan -- not process state."""

    def __init__(self, dims: list[int], acts: list[int]):
        self.flags = MLP_DEFAULT
        self._tls = threading.local()   # per-thread override of `flags` (`flags_override`): the plan is shared by threads
        self.L = len(acts)
        self.dims = (c_int * (self.L + 1))(*dims)
        self.acts = (c_int * self.L)(*acts)
        self._dims_list = list(dims)
        self._ws: dict[tuple[int, str], Tensor] = {}

    def effective_flags(self) -> int:
        """``flags`` OR-ed with the calling thread's override (none by default)."""
        return self.flags | getattr(self._tls, "extra", 0)

    @contextmanager
    def flags_override(self, extra: int):
        """Within the block, products issued by THIS thread carry ``flags | extra`` (an argument of their C calls); other
        threads using the same operator are not affected and nothing on the shared object is mutated."""
        prev = getattr(self._tls, "extra", 0)
        self._tls.extra = prev | int(extra)
        try:
            yield
        finally:
            self._tls.extra = prev

    def _ptr_array(self, tensors):
        arr = (c_void_p * self.L)()
        for i, t in enumerate(tensors):
            arr[i] = None if t is None else _pc(t)
        return arr

    def workspace(self, N: int, device) -> Tensor:
        key = (N, str(device))
        ws = self._ws.get(key)
        if ws is None:
            n = load().clo_mlp_ggn_ws_floats(self.L, self.dims, N)
            ws = torch.empty(n, device=device, dtype=torch.float32)
            # counters of the persistent <= 8-row kernel (zeroed once; the kernel maintains them)
            with torch.cuda.device(ws.device):
                rc = load().clo_mlp_ggn_ws_init(self.L, self.dims, N, ws.data_ptr(),
                                                torch.cuda.current_stream(ws.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"clo_mlp_ggn_ws_init failed: {load().clo_last_error().decode(errors='replace')}")
            self._ws = {k: v for k, v in self._ws.items() if k[0] in ("mm", "hess", "jac")}  # keep only the latest batch size
            self._ws[key] = ws
        return ws

    # ---- flat fast path: parameters / vectors addressed as base pointer + element offsets ----
    def bind_params(self, W, b) -> None:
        """Cache the (static) weight / bias pointer tables."""
        self._W_arr, self._b_arr = self._ptr_array(W), self._ptr_array(b)
        self._VW_arr, self._Vb_arr = (c_void_p * self.L)(), (c_void_p * self.L)()
        self._OW_arr, self._Ob_arr = (c_void_p * self.L)(), (c_void_p * self.L)()
        self._fn = load().clo_mlp_ggn_matvec

    def ggn_matvec_flat(self, v_base: int, o_base: int, w_off, b_off, X_ptr: int, N: int, loss_kind: int,
                        loss_scale: float, alpha: float, beta: float, aux_ptr, aux_rank: int, ws_ptr: int,
                        stream: int) -> None:
        """Like :meth:`ggn_matvec` for a flat vector / result: ``v_base``/``o_base`` are device
        addresses, ``w_off``/``b_off`` BYTE offsets of every layer's weight / bias (None = no bias).
        No tensor objects are created or inspected (host cost: a handful of integer adds)."""
        VW, Vb, OW, Ob = self._VW_arr, self._Vb_arr, self._OW_arr, self._Ob_arr
        for l in range(self.L):
            VW[l] = v_base + w_off[l]
            OW[l] = o_base + w_off[l]
            if b_off[l] is None:
                Vb[l] = None
                Ob[l] = None
            else:
                Vb[l] = v_base + b_off[l]
                Ob[l] = o_base + b_off[l]
        rc = self._fn(self.L, self.dims, self.acts, self._W_arr, self._b_arr, VW, Vb, OW, Ob, X_ptr, N,
                      loss_kind, aux_ptr, aux_rank, loss_scale, alpha, beta, self.effective_flags(), ws_ptr, stream)
        if rc != 0:
            _check(rc, "clo_mlp_ggn_matvec")

    # ---- K columns at once -------------------------------------------------------------------
    MATMAT_MAX_K = 64

    def matmat_supported(self, K: int, ldk: int, aux_rank: int = 1) -> bool:
        """Shape conditions of ``clo_mlp_ggn_matmat`` (pointer alignment is checked by the library)."""
        return (K % 4 == 0 and 4 <= K <= self.MATMAT_MAX_K and ldk % 4 == 0 and aux_rank <= 16
                and all(d % 4 == 0 for d in self._dims_list[:-1]))

    def matmat_workspace(self, K: int, device) -> Tensor:
        key = ("mm", K, str(device))
        ws = self._ws.get(key)
        if ws is None:
            n = load().clo_mlp_ggn_matmat_ws_floats(self.L, self.dims, 8, K)
            ws = torch.empty(n, device=device, dtype=torch.float32)
            self._ws[key] = ws
        return ws

    def ggn_matmat_ptrs(self, vw_ptrs, vb_ptrs, ow_ptrs, ob_ptrs, ldk: int, K: int, X_ptr: int, N: int,
                        loss_kind: int, loss_scale: float, alpha: float, beta: float, aux_ptr, aux_rank: int,
                        ws_ptr: int, stream: int) -> None:
        """``out[.., k] = beta out + alpha (J^T H J) V[.., k]`` for K columns; every ``*_ptrs`` entry is
        the device address of a layer's ``[d_out][d_in][K]`` (``[d_out][K]``) block with row stride
        ``ldk`` floats (None = no bias)."""
        VW, Vb, OW, Ob = self._VW_arr, self._Vb_arr, self._OW_arr, self._Ob_arr
        for l in range(self.L):
            VW[l], OW[l], Vb[l], Ob[l] = vw_ptrs[l], ow_ptrs[l], vb_ptrs[l], ob_ptrs[l]
        rc = load().clo_mlp_ggn_matmat(self.L, self.dims, self.acts, self._W_arr, self._b_arr, VW, Vb, OW, Ob,
                                       ldk, X_ptr, N, K, loss_kind, aux_ptr, aux_rank, loss_scale, alpha,
                                       beta, ws_ptr, stream)
        if rc != 0:
            _check(rc, "clo_mlp_ggn_matmat")

    def hessian_matmat_supported(self, K: int, ldk: int) -> bool:
        """Shape conditions of ``clo_mlp_hessian_matmat``: those of the GGN columns, one contiguous block of K
        columns, a linear last layer."""
        return self.matmat_supported(K, ldk) and ldk == K and int(self.acts[self.L - 1]) == 0

    def hessian_matmat_workspace(self, K: int, device) -> Tensor:
        key = ("hmm", K, str(device))
        ws = self._ws.get(key)
        if ws is None:
            n = load().clo_mlp_hessian_matmat_ws_floats(self.L, self.dims, 8, K)
            ws = torch.empty(n, device=device, dtype=torch.float32)
            self._ws[key] = ws
        return ws

    def hessian_matmat_ptrs(self, vw_ptrs, vb_ptrs, ow_ptrs, ob_ptrs, K: int, X_ptr: int, N: int, G_ptr: int,
                            loss_kind: int, loss_scale: float, alpha: float, beta: float, ws_ptr: int,
                            stream: int) -> None:
        """``out[.., k] = beta out + alpha H V[.., k]`` (exact Hessian) for K contiguous columns (``ldk == K``)."""
        VW, Vb, OW, Ob = self._VW_arr, self._Vb_arr, self._OW_arr, self._Ob_arr
        for l in range(self.L):
            VW[l], OW[l], Vb[l], Ob[l] = vw_ptrs[l], ow_ptrs[l], vb_ptrs[l], ob_ptrs[l]
        rc = load().clo_mlp_hessian_matmat(self.L, self.dims, self.acts, self._W_arr, self._b_arr, VW, Vb, OW, Ob,
                                           K, X_ptr, N, K, G_ptr, loss_kind, loss_scale, alpha, beta, ws_ptr, stream)
        if rc != 0:
            _check(rc, "clo_mlp_hessian_matmat")

    def _jac_workspace(self, N: int, device) -> Tensor:
        key = ("jac", N, str(device))
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(load().clo_mlp_jac_ws_floats(self.L, self.dims, N), device=device, dtype=torch.float32)
            self._ws = {k: v for k, v in self._ws.items() if k[0] != "jac"}
            self._ws[key] = ws
        return ws

    def jvp(self, W, b, VW, Vb, X, out: Tensor) -> None:
        """``out[n, c] = (J v)[n, c]`` for one mini-batch (``out`` contiguous ``[N, d_L]``)."""
        N = X.shape[0]
        rc = load().clo_mlp_jvp(self.L, self.dims, self.acts, self._ptr_array(W), self._ptr_array(b),
                                self._ptr_array(VW), self._ptr_array(Vb), _pc(X), N, _pc(out),
                                _pc(self._jac_workspace(N, X.device)), _stream())
        _check(rc, "clo_mlp_jvp")

    def vjp(self, W, b, OW, Ob, X, U: Tensor, alpha: float, beta: float) -> None:
        """``out = beta out + alpha J^T U`` for one mini-batch, ``U`` contiguous ``[N, d_L]``."""
        N = X.shape[0]
        rc = load().clo_mlp_vjp(self.L, self.dims, self.acts, self._ptr_array(W), self._ptr_array(b),
                                self._ptr_array(OW), self._ptr_array(Ob), _pc(X), N, _pc(U), alpha, beta,
                                _pc(self._jac_workspace(N, X.device)), _stream())
        _check(rc, "clo_mlp_vjp")

    def loss_grad(self, W, b, X: Tensor, targets: Tensor, loss_kind: int, scale: float) -> Tensor:
        """``G [N, C] = scale * d l_n / d f_n`` at ``f = net(X)`` from the live parameters (``clo_mlp_loss_grad``): the
        output gradient the exact-Hessian products take, without a forward / autograd pass on the host."""
        N = X.shape[0]
        G = torch.empty(N, self._dims_list[-1], device=X.device, dtype=torch.float32)
        rc = load().clo_mlp_loss_grad(self.L, self.dims, self.acts, self._ptr_array(W), self._ptr_array(b), _pc(X), N,
                                      loss_kind, _pc(targets), scale, _pc(G), _pc(self._jac_workspace(N, X.device)), _stream())
        _check(rc, "clo_mlp_loss_grad")
        return G

    def hessian_supported(self) -> bool:
        """Shape conditions of ``clo_mlp_hessian_matvec``: none any more (layer inputs that are not multiples
        of 4 run on the scalar-load kernel variants / the unaligned GEMM tile loader)."""
        return True

    def hessian_matvec(self, W, b, VW, Vb, OW, Ob, X, G, loss_kind: int, loss_scale: float, alpha: float,
                       beta: float, aux=None) -> None:
        """``out = beta out + alpha H v`` (exact Hessian, R-operator on the GEMM engine); ``G`` is the
        gradient of the reduced mini-batch loss w.r.t. the model output, ``[N, C]``."""
        lib = load()
        N = X.shape[0]
        key = ("hess", N, str(X.device))
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(lib.clo_mlp_hessian_ws_floats(self.L, self.dims, N), device=X.device,
                             dtype=torch.float32)
            self._ws = {k: v for k, v in self._ws.items() if k[0] != "hess"}
            self._ws[key] = ws
        rank = 1 if aux is None else (aux.shape[1] if aux.dim() == 3 else 1)
        rc = lib.clo_mlp_hessian_matvec(
            self.L, self.dims, self.acts, self._ptr_array(W), self._ptr_array(b),
            self._ptr_array(VW), self._ptr_array(Vb), self._ptr_array(OW), self._ptr_array(Ob),
            _pc(X), N, _pc(G), loss_kind, _pc(aux), rank, loss_scale, alpha, beta, _pc(ws), _stream(),
        )
        _check(rc, "clo_mlp_hessian_matvec")

    def ggn_matvec(self, W, b, VW, Vb, OW, Ob, X, loss_kind: int, loss_scale: float, alpha: float,
                   beta: float, aux=None) -> None:
        lib = load()
        N = X.shape[0]
        rank = 1 if aux is None else (aux.shape[1] if aux.dim() == 3 else 1)
        rc = lib.clo_mlp_ggn_matvec(
            self.L, self.dims, self.acts, self._ptr_array(W), self._ptr_array(b),
            self._ptr_array(VW), self._ptr_array(Vb), self._ptr_array(OW), self._ptr_array(Ob),
            _pc(X), N, loss_kind, _pc(aux), rank, loss_scale, alpha, beta, self.effective_flags(),
            _pc(self.workspace(N, X.device)), _stream(),
        )
        _check(rc, "clo_mlp_ggn_matvec")


def eigh_batched_(work: Tensor, n: int, max_blocks: int = 0) -> tuple[Tensor, Tensor]:
    """``clo_eigh_batched_f32``: eigendecomposition of the symmetric matrices in ``work[b, :n, :n]`` (fp32 GPU tensor
    ``[B, n, ld]``, ``ld % 4 == 0``, zero padding columns, contiguous; OVERWRITTEN) in one foreign call.  Returns
    ``(lam [B, n] ascending, Z [B, n, ld])`` with the eigenvectors in the ROWS of ``Z[b, :, :n]``."""
    lib = load()
    B, ld = work.shape[0], work.shape[2]
    lam = torch.empty(B, n, device=work.device, dtype=torch.float32)
    Z = torch.empty(B, n, ld, device=work.device, dtype=torch.float32)
    nbytes = lib.clo_eigh_ws_bytes(n, B)
    ws = torch.empty((nbytes + 3) // 4, device=work.device, dtype=torch.float32)
    with torch.cuda.device(work.device):
        rc = lib.clo_eigh_batched_f32(_p(work), ld, work.stride(0), n, B, _p(lam), n, _p(Z), ld, Z.stride(0), ws.data_ptr(),
                                      nbytes, max_blocks, torch.cuda.current_stream(work.device).cuda_stream)
    _check(rc, "clo_eigh_batched_f32")
    return lam, Z


def stedc_(d: Tensor, e: Tensor, n: int) -> tuple[Tensor, Tensor]:
    """``clo_stedc_f32``: ``d``, ``e`` fp32 GPU ``[B, >= n]`` (same row stride) -> ``(lam [B, n], Z [B, n, n])`` with the
    eigenvectors of the tridiagonal matrices in the ROWS of ``Z``."""
    lib = load()
    B = d.shape[0]
    if d.stride(0) != e.stride(0) or d.stride(1) != 1 or e.stride(1) != 1:
        d, e = d.contiguous(), e.contiguous()
        if d.shape != e.shape:
            e = torch.nn.functional.pad(e, (0, d.shape[1] - e.shape[1]))
    lam = torch.empty(B, n, device=d.device, dtype=torch.float32)
    Z = torch.empty(B, n, n, device=d.device, dtype=torch.float32)
    nbytes = lib.clo_stedc_ws_bytes(n, B)
    ws = torch.empty((nbytes + 3) // 4, device=d.device, dtype=torch.float32)
    with torch.cuda.device(d.device):
        rc = lib.clo_stedc_f32(_p(d), _p(e), d.stride(0), n, B, _p(lam), n, _p(Z), n, Z.stride(0), ws.data_ptr(), nbytes,
                               torch.cuda.current_stream(d.device).cuda_stream)
    _check(rc, "clo_stedc_f32")
    return lam, Z


# --------------------------------------------------------------------------------------
# streaming helpers
# --------------------------------------------------------------------------------------
def axpby(y: Tensor, x: Tensor, alpha: float, beta: float) -> Tensor:
    if y.shape != x.shape:
        raise ValueError("axpby shape mismatch")
    _check(load().clo_axpby_f32(_pc(y), _pc(x), y.numel(), alpha, beta, _stream()), "clo_axpby_f32")
    return y


def dot(x: Tensor, y: Tensor, scale: float = 1.0) -> Tensor:
    """``scale * <x, y>`` of two equally shaped contiguous fp32 GPU tensors as a 0-d tensor."""
    if x.shape != y.shape:
        raise ValueError("dot shape mismatch")
    lib = load()
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.clo_dot_ws_bytes(), device=x.device, dtype=torch.uint8)
    _check(lib.clo_dot_f32(_pc(x), _pc(y), x.numel(), scale, _pc(out), ws.data_ptr(), _stream()), "clo_dot_f32")
    return out[0]


def transpose(x: Tensor) -> Tensor:
    rows, cols = x.shape
    out = torch.empty(cols, rows, device=x.device, dtype=torch.float32)
    _check(load().clo_transpose_f32(_pc(out), _pc(x), rows, cols, _stream()), "clo_transpose_f32")
    return out


def canonical_pack(w: Tensor, bias: Tensor | None, rows: int, cols_w: int) -> Tensor:
    """K-trailing parameter tangents ``w [rows * cols_w, K]`` (+ ``bias [rows, K]``) -> K-major canonical
    ``[K, rows * (cols_w + 1)]`` (bias as the last column of every row) in one pass."""
    K = w.shape[-1]
    bc = cols_w + (bias is not None)
    out = torch.empty(K, rows * bc, device=w.device, dtype=torch.float32)
    _check(load().clo_canonical_pack_f32(_pc(out), _pc(w), _pc(bias) if bias is not None else None, rows, cols_w, K,
                                         _stream()), "clo_canonical_pack_f32")
    return out


def canonical_unpack(kmajor: Tensor, rows: int, cols_w: int, with_bias: bool) -> tuple[Tensor, Tensor | None]:
    """Inverse of :func:`canonical_pack`: ``[K, rows * (cols_w + with_bias)]`` -> ``(w [rows * cols_w, K], bias [rows, K])``."""
    K = kmajor.shape[0]
    w = torch.empty(rows * cols_w, K, device=kmajor.device, dtype=torch.float32)
    b = torch.empty(rows, K, device=kmajor.device, dtype=torch.float32) if with_bias else None
    _check(load().clo_canonical_unpack_f32(_pc(w), _pc(b) if b is not None else None, _pc(kmajor), rows, cols_w, K,
                                           _stream()), "clo_canonical_unpack_f32")
    return w, b


def rowscale(x: Tensor, s: Tensor, reciprocal: bool = False, shift: float = 0.0) -> Tensor:
    rows, K = x.shape
    y = torch.empty_like(x)
    _check(load().clo_rowscale_f32(_pc(y), _pc(x), _pc(s), rows, K, int(reciprocal), shift,
                                   _stream()), "clo_rowscale_f32")
    return y


def pack_probes(D: int, K: int, seed: int, distribution: str, device) -> Tensor:
    dist = {"rademacher": 0, "normal": 1}.get(distribution)
    if dist is None:
        raise ValueError(f"Unknown distribution {distribution!r}.")
    out = torch.empty(D, K, device=device, dtype=torch.float32)
    _check(load().clo_pack_probes_f32(_pc(out), D, K, seed & (2**64 - 1), dist, _stream()),
           "clo_pack_probes_f32")
    return out


_KRON_WS: dict = {}


def _kron_ws(device, floats: int) -> Tensor:
    """One growing workspace per (device, stream) for the Kronecker block calls (temporaries + split-K slabs)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _KRON_WS.get(key)
    if ws is None or ws.numel() < floats:
        ws = _KRON_WS[key] = torch.empty(floats, device=device, dtype=torch.float32)
    return ws


def _rm(S: Tensor) -> None:
    if S.dim() != 2 or (S.shape[1] > 1 and S.stride(1) != 1) or (S.shape[0] > 1 and S.stride(0) < S.shape[1]):
        raise ValueError("expected a row-major factor (unit column stride)")


def kron_blocks(blocks: list[tuple[Tensor, Tensor, Tensor | None, int]], xs: list[Tensor], K: int) -> list[Tensor]:
    """Every block of a block-diagonal Kronecker operator in ONE foreign call (``clo_kron_matmat_blocks``).

    ``blocks[i] = (S1 [A, a], S2 [B, b], lam | None, flags)`` (row-major fp32, any leading dimension; ``lam [A*B]`` marks
    an eigen-decomposed block ``(S1 (x) S2) diag(lam) (S1 (x) S2)^T``; bit 0 / 1 of ``flags``: array 1 / 2 holds the
    transposed factor resp. its eigenvectors in the rows), ``xs[i]`` the K-major operand, a contiguous ``[K, .]`` array.
    Returns the K-major results."""
    lib = load()
    n = len(blocks)
    dev = xs[0].device
    for S1, S2, _, _ in blocks:
        _rm(S1)
        _rm(S2)
    A = (c_int * n)(*[b[0].shape[0] for b in blocks])
    a = (c_int * n)(*[b[0].shape[1] for b in blocks])
    B = (c_int * n)(*[b[1].shape[0] for b in blocks])
    b_ = (c_int * n)(*[b[1].shape[1] for b in blocks])
    fl = (c_int * n)(*[int(b[3]) for b in blocks])
    ld1 = (c_long * n)(*[max(b[0].stride(0), b[0].shape[1]) for b in blocks])
    ld2 = (c_long * n)(*[max(b[1].stride(0), b[1].shape[1]) for b in blocks])
    ys, floats = [], 0
    for i, (S1, S2, lam, f) in enumerate(blocks):
        r1 = a[i] if (f & 1) and lam is None else A[i]
        r2 = b_[i] if (f & 2) and lam is None else B[i]
        ys.append(torch.empty(K, r1 * r2, device=dev, dtype=torch.float32))
        floats = max(floats, lib.clo_kron_ws_floats(A[i], a[i], B[i], b_[i], K, int(lam is not None)))
    ws = _kron_ws(dev, floats)
    ptrs = lambda ts: (c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts])   # noqa: E731
    _check(lib.clo_kron_matmat_blocks(n, ptrs(ys), ptrs([b[0] for b in blocks]), ld1, ptrs([b[1] for b in blocks]), ld2,
                                      ptrs([b[2] for b in blocks]), ptrs(xs), A, a, B, b_, fl, K, _pc(ws),
                                      ws.numel(), _stream()), "clo_kron_matmat_blocks")
    return ys


def kron_matmat(S1: Tensor, S2: Tensor, x: Tensor, K: int, trans: int = 0) -> Tensor:
    """``Y_k = E1 X_k E2^T`` for the K-major operand ``x``, ``E_i = S_i`` or ``S_i^T`` (bit i-1 of ``trans``):
    ``clo_kron_matmat``."""
    lib = load()
    _rm(S1)
    _rm(S2)
    A, a, B, b = S1.shape[0], S1.shape[1], S2.shape[0], S2.shape[1]
    y = torch.empty(K, (a if trans & 1 else A) * (b if trans & 2 else B), device=x.device, dtype=torch.float32)
    ws = _kron_ws(x.device, lib.clo_kron_ws_floats(A, a, B, b, K, 0))
    _check(lib.clo_kron_matmat(_pc(y), _p(S1), max(S1.stride(0), a), _p(S2), max(S2.stride(0), b), _pc(x), A, a, B, b, K,
                               int(trans), _pc(ws), ws.numel(), _stream()), "clo_kron_matmat")
    return y


def eigh_apply(Q1: Tensor, Q2: Tensor, lam: Tensor, x: Tensor, K: int, rows: int = 0) -> Tensor:
    """``Y_k = Q1 (lam .* (Q1^T X_k Q2)) Q2^T`` for the K-major operand ``x [K, n1*n2]``: ``clo_eigh_apply``
    (bit i-1 of ``rows``: array i holds its eigenvectors in the rows, i.e. ``Q_i^T``)."""
    lib = load()
    _rm(Q1)
    _rm(Q2)
    n1, n2 = Q1.shape[0], Q2.shape[0]
    y = torch.empty(K, n1 * n2, device=x.device, dtype=torch.float32)
    ws = _kron_ws(x.device, lib.clo_kron_ws_floats(n1, n1, n2, n2, K, 1))
    _check(lib.clo_eigh_apply(_pc(y), _p(Q1), max(Q1.stride(0), n1), _p(Q2), max(Q2.stride(0), n2), _pc(lam), _pc(x), n1, n2,
                              K, int(rows), _pc(ws), ws.numel(), _stream()), "clo_eigh_apply")
    return y


def ekfac_correction(g: Tensor, Qg: Tensor, a: Tensor, Qa: Tensor, rows: int = 0) -> Tensor:
    """``sum_{v,n} (Qg^T (sum_s g_vns a_ns^T) Qa)^2`` as ``[d_out, d_in]`` in ONE foreign call (``clo_ekfac_correction_f32``):
    ``g [V, B, S, d_out]``, ``a [B, S, d_in]`` contiguous; row-major eigenvector arrays, bit 0 / 1 of ``rows``: the array of
    ``Qg`` / ``Qa`` holds the eigenvectors in its rows."""
    lib = load()
    _rm(Qg)
    _rm(Qa)
    V, B, S, d1 = g.shape
    d2 = a.shape[-1]
    out = torch.empty(d1, d2, device=g.device, dtype=torch.float32)
    nws = lib.clo_ekfac_correction_ws_floats(V, B, S, d1, d2)
    ws = torch.empty(nws, device=g.device, dtype=torch.float32)
    _check(lib.clo_ekfac_correction_f32(_pc(out), d2, _p(Qg), max(Qg.stride(0), d1), _p(Qa), max(Qa.stride(0), d2), int(rows),
                                        _pc(g), _pc(a), V, B, S, d1, d2, 1.0, 0.0, _pc(ws), nws, _stream()),
           "clo_ekfac_correction_f32")
    return out
