"""Converters between parameter space and KFAC's canonical space.

Canonical form of a joint (W, b) group: ``W`` flattened to ``[d_out, d_in]``, the bias appended
as the LAST COLUMN, then flattened row-major; any other parameter is just flattened.  Blocks
follow the order of the parameter groups, inputs/outputs the order of the parameter dict
(reference ``curvlinops/kfac_utils.py:183-398``, tested with shuffled orders in
``test/test_kfac_utils.py:23-34``).
"""

from __future__ import annotations


import torch
from torch import Size, Tensor

from curvlinops_amd import _hip
from curvlinops_amd.linop import PyTorchLinearOperator
from curvlinops_amd.utils import is_native_tensor

# K-major column blocks go through the native pack / unpack kernels (tools/probe_canonical.py flips this attribute for A/B runs)
KMAJOR_BLOCKS = True


def is_kmajor(t: Tensor) -> bool:
    """``[n, K]`` tensor whose memory is the contiguous ``[K, n]`` array (K > 1): the layout the batched GEMMs of the
    Kronecker blocks read; the canonical converters and the blocks hand it to each other without transposes."""
    return t.ndim == 2 and t.shape[1] > 1 and t.shape[0] > 1 and t.stride(0) == 1 and t.stride(1) == t.shape[0]

ParamGroup = dict[str, str]  # role ("W" / "b") -> full parameter name


class _CanonicalBase(PyTorchLinearOperator):
    def __init__(self, param_shapes: dict[str, Size], param_groups: list[ParamGroup],
                 device: torch.device, dtype: torch.dtype):
        self._param_shapes = {k: Size(v) for k, v in param_shapes.items()}
        self._param_groups = param_groups
        self._device, self._dtype = device, dtype
        self._position = {name: i for i, name in enumerate(self._param_shapes)}
        param_space = [tuple(s) for s in self._param_shapes.values()]
        canon_space = self._canonical_shapes()
        if self._TO_CANONICAL:
            super().__init__(param_space, canon_space)
        else:
            super().__init__(canon_space, param_space)

    _TO_CANONICAL = True

    def _canonical_shapes(self) -> list[tuple[int, ...]]:
        out = []
        for g in self._param_groups:
            if "W" in g and "b" in g:
                w = self._param_shapes[g["W"]]
                out.append((w.numel() + w[0],))
            else:
                out.extend((self._param_shapes[n].numel(),) for n in g.values())
        return out

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return self._dtype


class ToCanonicalLinearOperator(_CanonicalBase):
    """Parameter space -> canonical space (``P^T`` in ``KFAC = P K P^T``)."""

    _TO_CANONICAL = True

    def _matmat(self, M: list[Tensor]) -> list[Tensor]:
        out = []
        for g in self._param_groups:
            if "W" in g and "b" in g:
                w = M[self._position[g["W"]]]
                b = M[self._position[g["b"]]]
                if (w.shape[-1] > 1 and is_native_tensor(w) and is_native_tensor(b) and w.is_contiguous()
                        and b.is_contiguous() and KMAJOR_BLOCKS):
                    # K columns on the GPU: bias column spliced in AND the K-major layout of the block's GEMMs in one
                    # pass (clo_canonical_pack_f32) instead of cat + transpose
                    rows, K = w.shape[0], w.shape[-1]
                    cols = w.numel() // (rows * K)
                    out.append(_hip.canonical_pack(w, b, rows, cols).T)
                    continue
                joint = torch.cat([w.flatten(start_dim=1, end_dim=-2), b.unsqueeze(1)], dim=1)
                out.append(joint.flatten(end_dim=-2))
            else:
                out.extend(M[self._position[n]].flatten(end_dim=-2) for n in g.values())
        return out

    def _adjoint(self) -> "FromCanonicalLinearOperator":
        return FromCanonicalLinearOperator(self._param_shapes, self._param_groups, self._device, self._dtype)


class FromCanonicalLinearOperator(_CanonicalBase):
    """Canonical space -> parameter space (``P``)."""

    _TO_CANONICAL = False

    def _matmat(self, M: list[Tensor]) -> list[Tensor]:
        out: list[Tensor | None] = [None] * len(self._param_shapes)
        (K,) = {m.shape[-1] for m in M}
        used = 0
        for g in self._param_groups:
            if "W" in g and "b" in g:
                w_shape = self._param_shapes[g["W"]]
                rows = w_shape[0]
                cols = w_shape.numel() // rows
                if is_kmajor(M[used]) and is_native_tensor(M[used]):   # K-major from the Kronecker block: one pass
                    w, b = _hip.canonical_unpack(M[used].T, rows, cols, True)
                    out[self._position[g["W"]]] = w.view(*w_shape, K)
                    out[self._position[g["b"]]] = b
                    used += 1
                    continue
                joint = M[used].reshape(rows, cols + 1, K)
                out[self._position[g["W"]]] = joint[:, :cols].reshape(*w_shape, K)
                out[self._position[g["b"]]] = joint[:, cols].reshape(rows, K)
                used += 1
            else:
                for n in g.values():
                    m = M[used]
                    if is_kmajor(m) and is_native_tensor(m):
                        m = _hip.canonical_unpack(m.T, m.shape[0], 1, False)[0]
                    out[self._position[n]] = m.reshape(*self._param_shapes[n], K)
                    used += 1
        if used != len(M) or any(o is None for o in out):
            raise RuntimeError("Mismatch in number of processed parameters.")
        return out

    def _adjoint(self) -> ToCanonicalLinearOperator:
        return ToCanonicalLinearOperator(self._param_shapes, self._param_groups, self._device, self._dtype)
