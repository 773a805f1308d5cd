"""The symmetric eigensolver behind ``clo_sytrd_f32`` without the vendor library (reference call sites:
``computers/_base.py:355-372``, ``kronecker.py:292-300`` -- ``torch.linalg.eigh`` = rocSOLVER ``ssyevd``):

* :func:`stedc_native`  -- eigen-decomposition of the symmetric TRIDIAGONAL matrix by Cuppen's divide & conquer
  (the algebra of LAPACK ``slaed1-4``): the matrix is torn into 2^k leaves of <= 64 rows, the leaves are solved by
  implicit QL (``clo_tql2_batched_f32``), and the tree is merged level by level with ALL nodes of a level in one
  batch: deflation scan, secular equation in float64, Gu-Eisenstat weights, eigenvector matrix of the rank-one
  update (``clo_dc_*``), and ONE batched GEMM pair ``Q_children @ M`` on the MFMA engine per level -- which is
  where the O(n^3) work is.  Clustered spectra (rank-deficient Kronecker factors) deflate, as in LAPACK.
* :func:`ormtr_native`  -- back-transformation with the Householder reflectors of the reduction as block
  reflectors ``I - V T V^T`` (64 per block, ``clo_larft_f32``): three GEMMs per block.

torch ops appear only as glue (sorts, gathers, index arithmetic on [nodes, s] arrays).
"""

from __future__ import annotations

import math

import torch
from torch import Tensor

from curvlinops_amd import _hip

_EPS32 = 2.0 ** -24
_LEAF = 64


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
    give ``([n], [n, n])``; a leading batch dimension ``[B, >= n]`` (matrices of one order: the factors of repeated
    layer shapes) gives ``([B, n], [B, n, n])`` with every tree level of ALL matrices in one batch."""
    single = d.dim() == 1
    d2 = d.reshape(1, -1) if single else d
    e2 = e.reshape(1, -1) if single else e
    lam, Q = _stedc_batched(d2, e2, n)
    return (lam[0], Q[0]) if single else (lam, Q)


def _stedc_batched(d: Tensor, e: Tensor, n: int) -> tuple[Tensor, Tensor]:
    dev = d.device
    B = d.shape[0]
    if n == 1:
        return d[:, :1].clone(), torch.ones(B, 1, 1, device=dev, dtype=torch.float32)
    k = max(0, math.ceil(math.log2(n / _LEAF)))
    L = -(-n // (1 << k))
    L = (L + 3) // 4 * 4            # multiples of 4 keep every block 16-byte aligned for the GEMM engine
    nleaf = 1 << k
    N = L * nleaf
    dp = torch.zeros(B, N, device=dev, dtype=torch.float64)
    ep = torch.zeros(B, N, device=dev, dtype=torch.float64)
    dp[:, :n] = d[:, :n].double()
    ep[:, : n - 1] = e[:, : n - 1].double()
    if N > n:  # decoupled padding: distinct values above the spectrum (they deflate in every merge)
        big = 4.0 * (dp[:, :n].abs().amax(dim=1) + 2.0 * ep.abs().amax(dim=1)) + 1.0
        dp[:, n:] = big[:, None] * (1.0 + 0.01 * torch.arange(1, N - n + 1, device=dev, dtype=torch.float64))[None, :]
    # tear at every leaf boundary c: T = diag(T1', T2') + beta (e_{c-1} + theta e_c)(...)^T with rho = |beta|
    cuts = torch.arange(1, nleaf, device=dev) * L
    beta = ep[:, cuts - 1].clone()
    dp[:, cuts - 1] -= beta.abs()
    dp[:, cuts] -= beta.abs()
    ep[:, cuts - 1] = 0.0
    beta_full = torch.zeros(B, N, device=dev, dtype=torch.float64)   # beta_full[:, c] = beta of the cut at row c
    beta_full[:, cuts] = beta
    # ---- leaves
    lam = torch.empty(B * nleaf, L, device=dev, dtype=torch.float32)
    Q = torch.empty(B * nleaf, L, L, device=dev, dtype=torch.float32)
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    d32, e32 = dp.float().contiguous(), ep.float().contiguous()
    # (float32 copies for the kernel interface; the tearing itself was done in float64)
    _run("clo_tql2_batched_f32", dev, _ptr(d32), _ptr(e32), _ptr(lam), _ptr(Q), L, B * nleaf, _ptr(status))
    lam = lam.double()
    # ---- merges, all nodes of a level (of all matrices) at once
    h = L
    while h < N:
        s = 2 * h
        per = N // s                   # nodes per matrix
        nodes = B * per
        Qc = Q.reshape(2 * nodes, h, h)
        starts = torch.arange(per, device=dev) * s
        b = beta_full[:, starts + h].reshape(nodes)
        theta = torch.where(b < 0, -torch.ones_like(b), torch.ones_like(b))
        rho = (2.0 * b.abs()).contiguous()
        z = torch.cat([Qc[0::2][:, h - 1, :].double(), theta[:, None] * Qc[1::2][:, 0, :].double()], dim=1) / math.sqrt(2.0)
        Ds, pi = torch.sort(lam.reshape(nodes, s), dim=1, stable=True)
        Ds = Ds.contiguous()
        zs = z.gather(1, pi).contiguous()
        typ = torch.empty(nodes, s, device=dev, dtype=torch.int32)
        rot_p = torch.empty(nodes, s, device=dev, dtype=torch.int32)
        rot_c = torch.empty(nodes, s, device=dev, dtype=torch.float64)
        rot_s = torch.empty(nodes, s, device=dev, dtype=torch.float64)
        K = torch.empty(nodes, device=dev, dtype=torch.int32)
        _run("clo_dc_deflate", dev, _ptr(Ds), _ptr(zs), _ptr(rho), _ptr(typ), _ptr(rot_p), _ptr(rot_c), _ptr(rot_s),
             _ptr(K), s, nodes, _EPS32)
        order = torch.sort(typ, dim=1, stable=True).indices           # survivors first, ascending position
        dk = Ds.gather(1, order).contiguous()
        zk = zs.gather(1, order).contiguous()
        spos = order.to(torch.int32).contiguous()
        kmax = s   # upper bound of the survivor counts: the kernels leave early per node (no host read of K)
        org = torch.zeros(nodes, s, device=dev, dtype=torch.int32)
        mu = torch.zeros(nodes, s, device=dev, dtype=torch.float64)
        zh = torch.zeros(nodes, s, device=dev, dtype=torch.float64)
        MT = torch.zeros(nodes, s, s, device=dev, dtype=torch.float32)
        _run("clo_dc_secular", dev, _ptr(dk), _ptr(zk), _ptr(rho), _ptr(K), _ptr(org), _ptr(mu), _ptr(zh), s, nodes, kmax)
        _run("clo_dc_build", dev, _ptr(dk), _ptr(K), _ptr(org), _ptr(mu), _ptr(zh), _ptr(spos), _ptr(MT), s, nodes, kmax)
        col = torch.arange(s, device=dev)[None, :].expand(nodes, s)
        defl = col >= K[:, None]                                       # columns K.. = deflated entries
        # unit entries of the deflated columns: MT[node, c, order[c]] += 1 for c >= K (adds 0 elsewhere; no mask
        # indexing, which would read the mask back on the host)
        MT.view(nodes, s * s).scatter_add_(1, col * s + order, defl.to(torch.float32))
        _run("clo_dc_rotate", dev, _ptr(MT), _ptr(rot_p), _ptr(rot_c), _ptr(rot_s), s, nodes)
        lam_u = torch.where(defl, dk, dk.gather(1, org.long()) + mu)
        lam_new, sigma = torch.sort(lam_u, dim=1, stable=True)
        # rows of M in ORIGINAL child order (undo the sort pi), columns in ascending-eigenvalue order (sigma)
        MpT = torch.empty_like(MT)
        MpT.scatter_(2, pi[:, None, :].expand(nodes, s, s), MT)
        MpT = MpT.gather(1, sigma[:, :, None].expand(nodes, s, s))
        Qn = torch.empty(nodes, s, s, device=dev, dtype=torch.float32)
        for half in (0, 1):
            _hip.gemm(Qc[half::2], MpT[:, :, half * h:(half + 1) * h].mT, out=Qn[:, half * h:(half + 1) * h, :])
        Q, lam, h = Qn, lam_new, s
    # a leaf that did not converge (status != 0; not observed) poisons the result instead of costing a host read
    # here: every caller verifies orthogonality / residual and falls back to float64
    poison = torch.where(status != 0, float("nan"), 0.0).to(torch.float32)[0]
    return lam.reshape(B, N)[:, :n].float() + poison, Q.reshape(B, N, N)[:, :n, :n]


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
