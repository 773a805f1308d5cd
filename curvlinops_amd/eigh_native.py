"""Argument marshalling for the hand-written symmetric eigensolver (reference call sites:
``computers/_base.py:355-372``, ``kronecker.py:292-300`` -- ``torch.linalg.eigh`` = rocSOLVER ``ssyevd`` there).

Since round 4 the whole solver is ONE foreign call (``clo_eigh_batched_f32``, ``csrc/eigh_driver.hip``): Householder
reduction in persistent panel launches, Cuppen's divide & conquer on the tridiagonal matrix with every level of the tree
walked in C++ (device-side sorts and gathers, the O(n^3) work as one batched GEMM pair per level), block-reflector
back-transformation.  This module only exposes the stages separately for the tests and probes:

* :func:`stedc_native`  -- the tridiagonal stage alone (``clo_stedc_f32``)
* :func:`ormtr_native`  -- back-transformation with the reflectors of the reduction (``clo_ormtr_f32``)
"""

from __future__ import annotations

import torch
from torch import Tensor

from curvlinops_amd import _hip



def _ptr(t: Tensor) -> int:
    return t.data_ptr()


def _run(name: str, dev: torch.device, *args) -> None:
    """Foreign call on the current stream of ``dev`` (pointers already converted)."""
    lib = _hip.load()
    with torch.cuda.device(dev):
        rc = getattr(lib, name)(*args, torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {lib.clo_last_error().decode(errors='replace')}")


# ----------------------------------------------------------------------------------------------
# tridiagonal divide & conquer
# ----------------------------------------------------------------------------------------------
def stedc_native(d: Tensor, e: Tensor, n: int) -> tuple[Tensor, Tensor]:
    """Eigenvalues (ascending, float32) and eigenvectors (in COLUMNS) of symmetric tridiagonal matrices with
    diagonal ``d[..., :n]`` and sub-diagonal ``e[..., :n-1]`` (float32 GPU tensors).  ``d``/``e`` of shape ``[>= n]``
    give ``([n], [n, n])``; a leading batch dimension ``[B, >= n]`` (matrices of one order) gives ``([B, n], [B, n, n])``
    with every tree level of ALL matrices in one batch.  One foreign call (``clo_stedc_f32``)."""
    single = d.dim() == 1
    d2 = (d.reshape(1, -1) if single else d).contiguous()
    e2 = (e.reshape(1, -1) if single else e).contiguous()
    if e2.shape[1] < d2.shape[1]:
        e2 = torch.nn.functional.pad(e2, (0, d2.shape[1] - e2.shape[1]))
    elif e2.shape[1] > d2.shape[1]:
        e2 = e2[:, : d2.shape[1]].contiguous()
    lam, Z = _hip.stedc_(d2, e2, n)
    Q = Z.mT   # eigenvectors in columns (a view: rows of Z)
    return (lam[0], Q[0]) if single else (lam, Q)


# ----------------------------------------------------------------------------------------------
# back-transformation
# ----------------------------------------------------------------------------------------------
def ormtr_native(work: Tensor, tau: Tensor, Zr: Tensor, n: int) -> None:
    """``Zr <- Zr Q^T`` in place, i.e. every ROW of ``Zr [m, ld]`` (an eigenvector of the tridiagonal matrix;
    ``ld >= pad4(n)``, multiple of 4, zero padding columns) is multiplied by ``Q = H_0 H_1 ... H_{n-2}``, the product
    of the Householder reflectors ``clo_sytrd_f32`` left in ``work`` (row i holds v_i in columns i+2.., unit entry at
    column i+1 implied) and ``tau``.  ONE foreign call (``clo_ormtr_f32``): blocks of 64 reflectors are applied as
    ``I - V T^T V^T`` in reverse order, three GEMMs per block on the MFMA engine."""
    if n < 3:
        return
    lib = _hip.load()
    m = Zr.shape[0]
    nws = lib.clo_ormtr_ws_floats(m, n)
    ws = torch.empty(nws, device=Zr.device, dtype=torch.float32)
    _run("clo_ormtr_f32", Zr.device, _ptr(work), work.stride(0), _ptr(tau), _ptr(Zr), Zr.stride(0), m, n, _ptr(ws), nws)
