"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy, float64) of the reference's
curvature-vector products for fully-connected nets.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``curvlinops_amd``) never does.

Pinned against the reference itself: ``oracle/make_golden.py`` imports f-dangel/curvlinops from
``/root/reference`` (this container only), evaluates its operators on seeded inputs and stores
inputs + outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function
below against those vectors.

Reference algorithm restated here (paths relative to the reference repository):
  * GGN-vector product  ``J^T (nabla_f^2 c) J v``: ``curvlinops/ggn.py:41-72``
  * Hessian-vector product (forward-over-reverse): ``curvlinops/hessian.py:13-69``
  * empirical-Fisher product as GGN of a pseudo-loss: ``curvlinops/gradient_moments.py:15-87``
  * data loop / normalisation ``sum_b (B_b / N_data) A_b v``: ``curvlinops/_torch_base.py:923-944``,
    ``curvlinops/_empirical_risk.py:340-352``
  * loss Hessians (MSE / CE / BCE): ``curvlinops/ggn_utils.py:29-85``
"""

from __future__ import annotations

import numpy as np

ACTS = ("identity", "relu", "tanh", "sigmoid")


def _act(name: str, z: np.ndarray):
    """Return (phi(z), phi'(z), phi''(z))."""
    if name == "identity":
        return z, np.ones_like(z), np.zeros_like(z)
    if name == "relu":
        m = (z > 0).astype(z.dtype)
        return z * m, m, np.zeros_like(z)
    if name == "tanh":
        t = np.tanh(z)
        return t, 1 - t * t, -2 * t * (1 - t * t)
    if name == "sigmoid":
        s = 1 / (1 + np.exp(-z))
        return s, s * (1 - s), s * (1 - s) * (1 - 2 * s)
    raise ValueError(name)


def forward(Ws, bs, acts, X):
    """Forward pass; returns lists (a_0..a_L, phi'_1..L, phi''_1..L)."""
    a = [np.asarray(X)]  # dtype follows the caller (float64 in tests, float32 for CPU timing)
    d1, d2 = [], []
    for W, b, act in zip(Ws, bs, acts):
        z = a[-1] @ W.T + (0 if b is None else b)
        out, p1, p2 = _act(act, z)
        a.append(out)
        d1.append(p1)
        d2.append(p2)
    return a, d1, d2


def _softmax(f):
    e = np.exp(f - f.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def reduction_factor(loss: str, reduction: str, N: int, C: int) -> float:
    """c such that the mini-batch loss is c * sum_n l_n (``ggn_utils.py:44-50`` and torch's
    ``reduction='mean'`` conventions: MSE/BCE average over N*C entries, CE over N terms)."""
    if reduction == "sum":
        return 1.0
    if reduction != "mean":
        raise ValueError(reduction)
    return 1.0 / N if loss == "ce" else 1.0 / (N * C)


def loss_value_grad(loss, reduction, f, y):
    """Mini-batch loss and its gradient w.r.t. the prediction f [N, C]."""
    N, C = f.shape
    c = reduction_factor(loss, reduction, N, C)
    if loss == "mse":
        return c * ((f - y) ** 2).sum(), 2 * c * (f - y)
    if loss == "ce":
        p = _softmax(f)
        onehot = np.eye(C)[y]
        return -c * np.log(p[np.arange(N), y]).sum(), c * (p - onehot)
    if loss == "bce":
        s = 1 / (1 + np.exp(-f))
        val = -(y * np.log(s) + (1 - y) * np.log(1 - s)).sum()
        return c * val, c * (s - y)
    raise ValueError(loss)


def loss_hessian_apply(loss, reduction, f, y, u):
    """(nabla_f^2 c)(f, y) u for the mini-batch loss, per sample (block-diagonal)."""
    N, C = f.shape
    c = reduction_factor(loss, reduction, N, C)
    if loss == "mse":
        return 2 * c * u
    if loss == "ce":
        p = _softmax(f)
        return c * (p * u - p * (p * u).sum(axis=1, keepdims=True))
    if loss == "bce":
        s = 1 / (1 + np.exp(-f))
        return c * s * (1 - s) * u
    raise ValueError(loss)


def _jvp(Ws, bs, acts, a, d1, vWs, vbs):
    """Forward-mode tangent of the output (``jvp(f)``, ggn.py:61); also per-layer tangents."""
    da = [np.zeros_like(a[0])]
    dzs = []
    for l, (W, vW) in enumerate(zip(Ws, vWs)):
        dz = da[-1] @ W.T + a[l] @ vW.T
        if vbs[l] is not None:
            dz = dz + vbs[l]
        dzs.append(dz)
        da.append(d1[l] * dz)
    return da, dzs


def _vjp(Ws, bs, a, d1, w_out):
    """Pull an output-space cotangent back to parameters (``vjp(f)``, ggn.py:68-71)."""
    L = len(Ws)
    gW, gb = [None] * L, [None] * L
    delta = w_out * d1[L - 1]
    for l in range(L - 1, -1, -1):
        gW[l] = delta.T @ a[l]
        gb[l] = None if bs[l] is None else delta.sum(axis=0)
        if l > 0:
            delta = (delta @ Ws[l]) * d1[l - 1]
    return gW, gb


def jacobian_matvec_batch(Ws, bs, acts, X, vWs, vbs):
    """Mini-batch Jacobian applied to a parameter-space vector: ``[N, C]`` (jacobian.py:33-51)."""
    a, d1, _ = forward(Ws, bs, acts, X)
    da, _ = _jvp(Ws, bs, acts, a, d1, vWs, vbs)
    return da[-1]


def jacobian_t_matvec_batch(Ws, bs, acts, X, U):
    """Transposed mini-batch Jacobian applied to an output-space vector ``U [N, C]``
    (jacobian.py:77-98): parameter-shaped results."""
    a, d1, _ = forward(Ws, bs, acts, X)
    return _vjp(Ws, bs, a, d1, U)


def ggn_matvec_batch(Ws, bs, acts, X, y, loss, reduction, vWs, vbs):
    """Mini-batch GGN-vector product (ggn.py:41-72)."""
    a, d1, _ = forward(Ws, bs, acts, X)
    da, _ = _jvp(Ws, bs, acts, a, d1, vWs, vbs)
    w = loss_hessian_apply(loss, reduction, a[-1], y, da[-1])
    return _vjp(Ws, bs, a, d1, w)


def ef_matvec_batch(Ws, bs, acts, X, y, loss, reduction, vWs, vbs):
    """Mini-batch empirical-Fisher product = GGN of 0.5/c sum_n <f_n, c g_n>^2
    (gradient_moments.py:48-87); g_n detached per-sample gradient of the mini-batch loss."""
    a, d1, _ = forward(Ws, bs, acts, X)
    N, C = a[-1].shape
    _, g = loss_value_grad(loss, reduction, a[-1], y)
    c = {"sum": 1.0, "mean": float(N if loss == "ce" else N * C)}[reduction]
    g = g * c
    da, _ = _jvp(Ws, bs, acts, a, d1, vWs, vbs)
    w = (1.0 / c) * g * (g * da[-1]).sum(axis=1, keepdims=True)
    return _vjp(Ws, bs, a, d1, w)


def mc_ggn_matvec_batch(Ws, bs, acts, X, reduction, grad_samples, vWs, vbs):
    """MC-GGN product for GIVEN sampled output gradients g'[N, M, C] (already scaled by
    1/sqrt(M)); pseudo-loss 0.5/c sum_{n,k} <g'_nk, f_n>^2 (ggn.py:140-166)."""
    a, d1, _ = forward(Ws, bs, acts, X)
    N = a[-1].shape[0]
    c = {"mean": float(N), "sum": 1.0}[reduction]
    da, _ = _jvp(Ws, bs, acts, a, d1, vWs, vbs)
    ip = np.einsum("nkc,nc->nk", grad_samples, da[-1])
    w = (1.0 / c) * np.einsum("nkc,nk->nc", grad_samples, ip)
    return _vjp(Ws, bs, a, d1, w)


def hessian_matvec_batch(Ws, bs, acts, X, y, loss, reduction, vWs, vbs):
    """Mini-batch Hessian-vector product, Pearlmutter R-operator == ``jvp(jacrev(loss))``
    (hessian.py:66)."""
    L = len(Ws)
    a, d1, d2 = forward(Ws, bs, acts, X)
    da, dzs = _jvp(Ws, bs, acts, a, d1, vWs, vbs)
    _, g = loss_value_grad(loss, reduction, a[-1], y)
    Rg = loss_hessian_apply(loss, reduction, a[-1], y, da[-1])
    # backward with tangents: delta_l = dL/dz_l, Rdelta_l its directional derivative
    delta = g * d1[L - 1]
    Rdelta = Rg * d1[L - 1] + g * d2[L - 1] * dzs[L - 1]
    hW, hb = [None] * L, [None] * L
    for l in range(L - 1, -1, -1):
        hW[l] = Rdelta.T @ a[l] + delta.T @ da[l]
        hb[l] = None if bs[l] is None else Rdelta.sum(axis=0)
        if l > 0:
            back = delta @ Ws[l]
            Rback = Rdelta @ Ws[l] + delta @ vWs[l]
            Rdelta = Rback * d1[l - 1] + back * d2[l - 1] * dzs[l - 1]
            delta = back * d1[l - 1]
    return hW, hb


_BATCH_FNS = {
    "ggn": ggn_matvec_batch,
    "ef": ef_matvec_batch,
    "hessian": hessian_matvec_batch,
}


def matvec(kind, Ws, bs, acts, data, loss, reduction, vWs, vbs, num_data=None, dtype=np.float64):
    """Whole-data-set product: sum over mini-batches with the reference's normalisation
    (``_torch_base.py:937-942``: factor 1 for 'sum', B_b / N_data for 'mean')."""
    Ws = [np.asarray(W, dtype=dtype) for W in Ws]
    bs = [None if b is None else np.asarray(b, dtype=dtype) for b in bs]
    vWs = [np.asarray(v, dtype=dtype) for v in vWs]
    vbs = [None if v is None else np.asarray(v, dtype=dtype) for v in vbs]
    if num_data is None:
        num_data = sum(X.shape[0] for X, _ in data)
    oW = [np.zeros_like(W) for W in Ws]
    ob = [None if b is None else np.zeros_like(b) for b in bs]
    fn = _BATCH_FNS[kind]
    for X, y in data:
        X = np.asarray(X, dtype=dtype)
        y = np.asarray(y)
        if loss != "ce":
            y = y.astype(dtype)
        norm = 1.0 if reduction == "sum" else X.shape[0] / num_data
        gW, gb = fn(Ws, bs, acts, X, y, loss, reduction, vWs, vbs)
        for l in range(len(Ws)):
            oW[l] += norm * gW[l]
            if ob[l] is not None:
                ob[l] += norm * gb[l]
    return oW, ob


def flatten_params(Ws, bs):
    """Concatenate in ``nn.Sequential`` parameter order (W_1, b_1, W_2, b_2, ...)."""
    parts = []
    for W, b in zip(Ws, bs):
        parts.append(np.asarray(W).reshape(-1))
        if b is not None:
            parts.append(np.asarray(b).reshape(-1))
    return np.concatenate(parts)


def unflatten_params(vec, shapes_W, has_bias):
    """Inverse of :func:`flatten_params`."""
    Ws, bs, off = [], [], 0
    for (do, di), hb in zip(shapes_W, has_bias):
        Ws.append(np.asarray(vec[off : off + do * di]).reshape(do, di))
        off += do * di
        if hb:
            bs.append(np.asarray(vec[off : off + do]))
            off += do
        else:
            bs.append(None)
    return Ws, bs
