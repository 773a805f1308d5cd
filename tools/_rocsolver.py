"""The two rocSOLVER routines the symmetric eigensolver keeps from the vendor library, through its C
API (ctypes on the copy PyTorch ships, so no second rocBLAS is loaded):

* ``sstedc``  -- divide & conquer eigensolver of the TRIDIAGONAL matrix (5.7 ms at n = 4609), and
* ``sormtr``  -- multiplication by the Householder reflectors of the reduction (12.9 ms).

The reduction itself (85 % of ``torch.linalg.eigh``'s time) is ``clo_sytrd_f32``.  One rocBLAS handle
per (device, HIP stream), kept for the life of the process.
"""

from __future__ import annotations

import ctypes
import os
import threading

import torch

_P, _I = ctypes.c_void_p, ctypes.c_int
_FILL_LOWER, _SIDE_LEFT, _OP_NONE, _EVECT_TRIDIAGONAL = 122, 141, 111, 212
_libs = None
_lock = threading.Lock()


def _load():
    global _libs
    if _libs is None:
        with _lock:
            if _libs is None:
                lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
                rocblas = ctypes.CDLL(os.path.join(lib_dir, "librocblas.so"), mode=ctypes.RTLD_GLOBAL)
                rocsolver = ctypes.CDLL(os.path.join(lib_dir, "librocsolver.so"), mode=ctypes.RTLD_GLOBAL)
                rocblas.rocblas_create_handle.argtypes = [ctypes.POINTER(_P)]
                rocblas.rocblas_set_stream.argtypes = [_P, _P]
                rocsolver.rocsolver_sstedc.argtypes = [_P, _I, _I, _P, _P, _P, _I, _P]
                rocsolver.rocsolver_sormtr.argtypes = [_P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I]
                _libs = (rocblas, rocsolver)
    return _libs


def available() -> bool:
    try:
        _load()
        return True
    except OSError:
        return False


_handles: dict = {}   # (device index, HIP stream) -> rocBLAS handle, process-wide


def _handle(device: torch.device):
    """One rocBLAS handle per (device, HIP stream), created once and kept for the life of the process.
    The worker threads of ``linalg_native`` are started per call but their HIP streams persist, so keying the
    handles by stream (not by thread, as rounds 1-2 did) bounds their number -- and the >= 32 MB of device
    workspace each one owns -- by the number of streams instead of leaking one per worker thread per
    refresh.  Work of one stream is ordered by the stream, so its handle (and workspace) is never used by
    two kernels at once."""
    rocblas, _ = _load()
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(device).cuda_stream
    key = (idx, stream)
    with _lock:
        h = _handles.get(key)
        if h is None:
            h = _P()
            with torch.cuda.device(idx):
                status = rocblas.rocblas_create_handle(ctypes.byref(h))
            if status != 0:
                raise RuntimeError(f"rocblas_create_handle failed with status {status}")
            status = rocblas.rocblas_set_stream(h, _P(stream))
            if status != 0:
                raise RuntimeError(f"rocblas_set_stream failed with status {status}")
            _handles[key] = h
    return h


def stedc_(D: torch.Tensor, E: torch.Tensor, Z: torch.Tensor, n: int) -> torch.Tensor:
    """Eigen-decomposition of the symmetric tridiagonal matrix (D, E): on return ``D`` holds the
    eigenvalues in ascending order and the column-major ``Z`` (``Z.stride(0)`` = leading dimension) the
    eigenvectors.  Returns the device ``info`` tensor (0 = converged)."""
    _, rocsolver = _load()
    info = torch.zeros(1, dtype=torch.int32, device=D.device)
    status = rocsolver.rocsolver_sstedc(_handle(D.device), _EVECT_TRIDIAGONAL, n, D.data_ptr(), E.data_ptr(),
                                        Z.data_ptr(), Z.stride(0), info.data_ptr())
    if status != 0:
        raise RuntimeError(f"rocsolver_sstedc failed with status {status}")
    return info


def ormtr_(A: torch.Tensor, tau: torch.Tensor, Z: torch.Tensor, n: int) -> None:
    """``Z <- Q Z`` with Q the product of the Householder reflectors stored (LAPACK ``uplo='L'``,
    column-major) in ``A`` / ``tau``; ``Z`` column-major with leading dimension ``Z.stride(0)``."""
    _, rocsolver = _load()
    status = rocsolver.rocsolver_sormtr(_handle(A.device), _SIDE_LEFT, _FILL_LOWER, _OP_NONE, n, n, A.data_ptr(),
                                        A.stride(0), tau.data_ptr(), Z.data_ptr(), Z.stride(0))
    if status != 0:
        raise RuntimeError(f"rocsolver_sormtr failed with status {status}")
