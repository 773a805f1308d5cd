"""Kronecker-factor computers: the ``backend`` plug-in point of ``KFACLinearOperator`` /
``EKFACLinearOperator`` (reference ``curvlinops/kfac.py:89-92``, ``computers/_base.py``).

``HipKFACComputer.compute()`` returns ``(input_covariances, gradient_covariances, mapping)`` and
``HipEKFACComputer.compute()`` the 4-tuple ``(Q_a, Q_g, corrected_eigenvalues, mapping)``, exactly
the contract of ``computers/_base.py:181-197, 305-327``.  Layer inputs and output-gradients are
harvested with module hooks during ordinary PyTorch forward/backward passes (host framework);
every contraction on them -- ``A += x^T x / (N S)``, ``G += corr * g^T g``
(``computers/kfac_hooks.py:350,390``), the eigenvalue correction
(``computers/ekfac_hooks.py:25-238``), the eigendecompositions -- runs on the HIP library when the
tensors are fp32 on the GPU.
"""

from __future__ import annotations

import os


from collections.abc import Callable, Iterable, MutableMapping
from contextlib import contextmanager
from functools import partial
from typing import Any

import torch
from torch import Tensor
from torch.func import vmap
from torch.nn import BCEWithLogitsLoss, Conv2d, CrossEntropyLoss, Linear, Module, MSELoss
from torch.nn.functional import unfold
from torch.nn.modules.utils import _pair

from curvlinops_amd import _hip, linalg_native
from curvlinops_amd.canonical import ParamGroup
from curvlinops_amd.enums import FisherType, KFACType
from curvlinops_amd.loss_sampling import make_grad_output_fn
from curvlinops_amd.risk import EmpiricalRiskMixin
from curvlinops_amd.utils import flatten_output_and_labels, is_native_tensor, seed_generator, side_stream

ParamGroupKey = tuple[str, ...]


# ------------------------------------------------------------------------------------------
# weight-sharing formats (reference computers/kfac_math.py, kfac_utils.py:78-180)
# ------------------------------------------------------------------------------------------
def _conv_hyperparams(mod: Module) -> dict[str, Any]:
    if isinstance(mod, Conv2d):
        return dict(kernel_size=mod.kernel_size, stride=mod.stride, padding=mod.padding,
                    dilation=mod.dilation, groups=mod.groups)
    return {}


def _string_padding(kernel_size: int, padding: str, dilation: int) -> tuple[int, int]:
    """(left, right) zero padding of ``padding='valid'|'same'`` (einconv.utils.get_conv_paddings)."""
    if padding == "valid":
        return 0, 0
    if padding == "same":
        total = dilation * (kernel_size - 1)
        return total // 2, total - total // 2
    raise ValueError(f"Unknown string padding {padding!r}.")


def _index_pattern(input_size: int, kernel_size: int, stride: int, padding, dilation: int,
                   device, dtype) -> Tensor:
    """``pattern[k, o, i] = (i == o*stride + k*dilation - pad_left)`` -- the connectivity of a
    1-d convolution (einconv ``index_pattern``; pinned by the reference's KFAC-reduce tests)."""
    left, right = _string_padding(kernel_size, padding, dilation) if isinstance(padding, str) else (padding, padding)
    out = (input_size + left + right - dilation * (kernel_size - 1) - 1) // stride + 1
    k = torch.arange(kernel_size, device=device).view(-1, 1, 1)
    o = torch.arange(out, device=device).view(1, -1, 1)
    i = torch.arange(input_size, device=device).view(1, 1, -1)
    return (i == o * stride + k * dilation - left).to(dtype)


def _group_mean(x: Tensor, groups: int) -> Tensor:
    if groups == 1:
        return x
    b, c, h, w = x.shape
    return x.reshape(b, groups, c // groups, h, w).mean(dim=1)


def extract_patches(x: Tensor, kernel_size, stride, padding, dilation, groups: int) -> Tensor:
    """im2col: ``[B, C, I1, I2] -> [B, O1*O2, C/groups * K1*K2]`` (groups averaged)."""
    if isinstance(padding, str):
        pads = []
        for k, d in zip(_pair(kernel_size), _pair(dilation)):
            left, right = _string_padding(k, padding, d)
            if left != right:
                raise NotImplementedError("Unequal padding not supported in unfold.")
            pads.append(left)
        padding = tuple(pads)
    x = _group_mean(x, groups)
    if is_native_tensor(x):  # one HIP launch for the whole batch (torch unfold: one per sample)
        return _hip.im2col(x, _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation))
    cols = unfold(x, kernel_size, dilation=dilation, padding=padding, stride=stride)
    return cols.transpose(1, 2)


def extract_averaged_patches(x: Tensor, kernel_size, stride, padding, dilation, groups: int) -> Tensor:
    """KFAC-reduce for convolutions: patches averaged over output positions,
    ``[B, C, I1, I2] -> [B, C/groups * K1*K2]``."""
    x = _group_mean(x, groups)
    pats = []
    pad_pair = (padding, padding) if isinstance(padding, str) else _pair(padding)
    for size, k, s, p, d in zip(x.shape[-2:], _pair(kernel_size), _pair(stride), pad_pair, _pair(dilation)):
        pats.append(_index_pattern(size, k, s, p, d, x.device, x.dtype).mean(dim=1))  # [k, i]
    out = torch.einsum("bcij,ki,lj->bckl", x, pats[0], pats[1])
    return out.flatten(1)


def input_to_weight_sharing_format(x: Tensor, kfac_approx: str, hyper: dict | None = None) -> Tensor:
    """Layer input -> ``[batch, shared, d_in]`` (``shared = 1`` for KFAC-reduce / plain Linear)."""
    if hyper:
        fn = extract_patches if kfac_approx == KFACType.EXPAND else extract_averaged_patches
        x = fn(x, hyper["kernel_size"], hyper["stride"], hyper["padding"], hyper["dilation"], hyper["groups"])
    if x.ndim == 2:
        return x.unsqueeze(1)
    if kfac_approx == KFACType.REDUCE:
        return x.flatten(1, -2).mean(dim=1, keepdim=True)
    return x.flatten(1, -2)


def grad_to_weight_sharing_format(g: Tensor, kfac_approx: str, hyper: dict | None = None) -> Tensor:
    """Output gradient -> ``[batch, shared, d_out]`` (KFAC-reduce SUMS over shared positions)."""
    if hyper:
        g = g.movedim(1, -1)
    if g.ndim == 2:
        return g.unsqueeze(1)
    if kfac_approx == KFACType.REDUCE:
        return g.flatten(1, -2).sum(dim=1, keepdim=True)
    return g.flatten(1, -2)


def compute_loss_correction(batch_size: int, terms_per_datum: int, reduction: str, n_data: int | None) -> float:
    """Undo the mean-reduction scaling that was baked into the backpropagated vectors:
    ``(B T)^2 / (T N_data)`` for 'mean', 1 for 'sum' (``computers/kfac_math.py:172-203``)."""
    if reduction == "sum":
        return 1.0
    denom = terms_per_datum * (n_data if n_data is not None else 1)
    return (batch_size * terms_per_datum) ** 2 / denom


@contextmanager
def _use_params(module: Module, params: dict[str, Tensor]):
    """Temporarily point the module's Parameters at the tensors in ``params`` (and run eval-mode BatchNorm layers
    as the affine maps they are, see `_affine_eval_batchnorm`)."""
    saved = []
    for name, value in params.items():   # (by name: walking named_parameters() twice cost 0.6 ms per ResNet-18 build)
        try:
            p = module.get_parameter(name)
        except AttributeError:
            continue
        saved.append((p, p.data))
        p.data = value
    patched = _affine_eval_batchnorm(module, params)
    try:
        yield
    finally:
        for mod in patched:
            del mod.forward
        for p, data in saved:
            p.data = data


_FAST_BN = True      # (module attributes: the A/B scripts under tools/ set them before building a computer)


def _affine_eval_batchnorm(module: Module, params: dict[str, Tensor]) -> list[Module]:
    """BatchNorm in eval mode is a per-channel affine map.  On this stack it dispatches to MIOpen's
    ``BatchNormFwdInferSpatialEst`` kernel, which takes ~200 us per layer on CIFAR-sized ResNet-18 batches
    (rocprofv3, profiles/r03_kfac_resnet18_build_kernels.txt: 20 calls = 4.0 ms, a third of all kernel time of a
    factor build).  For the duration of the KFAC passes every eval-mode BatchNorm with running statistics is
    evaluated as ``x * scale + shift`` (ONE fused elementwise launch, same autograd semantics w.r.t. its input;
    BatchNorm parameters get no gradient here, which no KFAC pass asks for);
    BatchNorm parameters are never among the Kronecker-factored ones (`kfac_hooks.py:445-449`), so this changes
    nothing but the rounding of the activations (~1e-7 relative).  ``computers._FAST_BN = False`` keeps the stock
    kernels.  Returns the modules whose ``forward`` was shadowed."""
    if not _FAST_BN or not isinstance(module, Module):
        return []
    tracked = {id(p) for p in params.values()}
    mods = []
    for mod in module.modules():
        if not isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) or mod.training:
            continue
        if mod.running_mean is None or mod.running_var is None or not mod.running_mean.is_cuda:
            continue
        if any(id(p) in tracked for p in mod.parameters(recurse=False)) or "forward" in mod.__dict__:
            continue
        mods.append(mod)
    if not mods:
        return []
    # scale / shift are recomputed on every entry (a cache across calls keyed by tensor identity / version went stale under
    # `.data` updates, which bump neither) -- for ALL layers at once with multi-tensor ops: a handful of launches per entry
    # instead of four per layer (ResNet-18: 160 tiny launches = 1.2 ms of host time per factor build)
    with torch.no_grad():
        inv = torch._foreach_add([m.running_var for m in mods], [float(m.eps) for m in mods])
        torch._foreach_sqrt_(inv)
        torch._foreach_reciprocal_(inv)
        with_w = [i for i, m in enumerate(mods) if m.weight is not None]
        if with_w:
            torch._foreach_mul_([inv[i] for i in with_w], [mods[i].weight for i in with_w])
        scales = inv
        shifts = torch._foreach_mul([m.running_mean for m in mods], scales)
        torch._foreach_neg_(shifts)
        with_b = [i for i, m in enumerate(mods) if m.bias is not None]
        if with_b:
            torch._foreach_add_([shifts[i] for i in with_b], [mods[i].bias for i in with_b])

    def make_forward(scale, shift):
        def forward(x):
            shape = (1, -1) + (1,) * (x.dim() - 2)
            return torch.addcmul(shift.view(shape), x, scale.view(shape))

        return forward

    for mod, scale, shift in zip(mods, scales, shifts):
        mod.forward = make_forward(scale, shift)
    return mods


# Factor accumulation (im2col + SYRK) runs on its own HIP stream so that it overlaps the autograd
# kernels of the layers that follow: the hook only orders it after the producer of its operand.
_FACTOR_STREAMS: dict = {}
_FACTOR_EVENTS: dict = {}
_OVERLAP = True


class _factor_stream:
    """``with _factor_stream(t):`` -- on fp32 GPU tensors, run the body on the per-device factor
    stream once ``t`` (produced on the current stream) is ready; no-op elsewhere."""

    def __init__(self, t: Tensor, on: bool = True):
        self._on = on and _OVERLAP and is_native_tensor(t)
        self._t = t

    def __enter__(self):
        if not self._on:
            return self
        dev = self._t.device
        side = _FACTOR_STREAMS.get(dev)
        if side is None:
            side = _FACTOR_STREAMS[dev] = side_stream(dev, 0)   # (first of the package-wide worker streams)
        ev = _FACTOR_EVENTS.get(dev)
        if ev is None:   # one reusable event per device (a wait captures the record that precedes it)
            ev = _FACTOR_EVENTS[dev] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        side.wait_event(ev)
        self._t.record_stream(side)
        self._ctx = torch.cuda.stream(side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._on:
            self._ctx.__exit__(*exc)
        return False


def _join_factor_stream(device) -> None:
    """Make the current stream wait for everything queued on the factor stream."""
    side = _FACTOR_STREAMS.get(device)
    if side is not None:
        torch.cuda.current_stream(device).wait_stream(side)


class _FactorStore(dict):
    """``{group key: factor}``.  With ``preallocate`` every factor is a view into ONE flat buffer
    (``.flat``) that the data-parallel build all-reduces in place -- no pack / unpack copies around
    the collective; ``fresh`` holds the keys whose view has not been written yet."""

    def __init__(self):
        super().__init__()
        self.flat: Tensor | None = None
        self.fresh: set = set()

    _ALIGN = 64   # floats: every factor starts on a 256-byte boundary

    def preallocate(self, sizes: dict, device, dtype) -> None:
        """``sizes``: ``{key: d}`` for ``d x d`` factors, laid out in dict order, each on a 256-byte boundary: joint
        weight + bias factors have odd orders (577, 1153, ...), and a factor that starts at an odd float offset falls off
        the 16-byte-aligned paths of every kernel that touches it later (Kronecker products 1.3 -> 1.9 ms, damped
        inverses 11.7 -> 16.8 ms on ResNet-18).  The few padding floats between factors are never read."""
        self.offsets, off = {}, 0
        for key, d in sizes.items():
            self.offsets[key] = off
            off += -(-(d * d) // self._ALIGN) * self._ALIGN
        self.flat = torch.empty(off, device=device, dtype=dtype)
        for key, d in sizes.items():
            o = self.offsets[key]
            self[key] = self.flat[o:o + d * d].view(d, d)
        self.fresh = set(sizes)

    def end_of(self, keys) -> int:
        """Flat offset just behind the last of ``keys`` (incl. its padding): ``flat[:end_of(keys)]`` holds them all."""
        ends = [self.offsets[k] + -(-self[k].numel() // self._ALIGN) * self._ALIGN for k in keys]
        return max(ends) if ends else 0


def _gram_accumulate(store: dict, key, X2d: Tensor, alpha: float, ones_col: bool) -> None:
    """``store[key] += alpha * [X|1]^T [X|1]`` -- HIP SYRK for fp32 GPU tensors."""
    d = X2d.shape[1] + (1 if ones_col else 0)
    fresh = getattr(store, "fresh", None)
    if is_native_tensor(X2d):
        X2d = X2d if X2d.stride(-1) == 1 else X2d.contiguous()
        C = store.get(key)
        first = C is None or (fresh is not None and key in fresh)
        if C is None:
            C = torch.empty(d, d, device=X2d.device, dtype=torch.float32)
            store[key] = C
        if fresh:
            fresh.discard(key)
        _hip.syrk_accum(C, X2d, alpha=alpha, beta=0.0 if first else 1.0, ones_col=ones_col)
        return
    if ones_col:
        X2d = torch.cat([X2d, X2d.new_ones(X2d.shape[0], 1)], dim=1)
    upd = (X2d.T @ X2d).mul_(alpha)
    if fresh is not None and key in fresh:
        fresh.discard(key)
        store[key].copy_(upd)
    elif key in store:
        store[key].add_(upd)
    else:
        store[key] = upd


# Conv2d input covariances: "1" = patches generated inside the SYRK's tile loader wherever that is the
# faster or the only memory-friendly way (default), "0" = always materialise them, "all" = always fused.
_FUSED_IM2COL = "1"   # "0": always materialise the patch matrix, "all": never
_PATCH_LIMIT_BYTES = 1024 * 2**20


def _conv_patches_fusable(hyper: dict) -> bool:
    pad = hyper["padding"]
    if isinstance(pad, str):
        try:
            pads = [_string_padding(k, pad, d) for k, d in zip(_pair(hyper["kernel_size"]), _pair(hyper["dilation"]))]
        except Exception:  # noqa: BLE001
            return False
        return all(l == r for l, r in pads)
    return True


def _use_fused_patches(hyper: dict, x: Tensor, ones_col: bool) -> bool:
    """Measured on MI355X (`tools/probe_im2col_syrk.py`, profiles/r02_im2col_syrk_fused.txt): the gather
    loader costs the symmetric GEMM 20-40 % of its MFMA rate, the materialised patches only their write +
    read (75 MB: 30 us of a 360 us SYRK).  So the fused kernel runs where the materialised path is worse
    off: patch lengths that are not float4-complete (ResNet stem 3x7x7 = 147: 231 vs 440 us; LeNet conv2
    6x5x5 = 150: 112 vs 198 us), which would fall off the aligned engine, and patch matrices beyond
    `CLO_KFAC_PATCH_LIMIT_MB` (default 1 GiB), where the copy is a memory problem before it is a time one."""
    if _FUSED_IM2COL == "0" or not _conv_patches_fusable(hyper):
        return False
    if _FUSED_IM2COL == "all":
        return True
    ks, st, dl = _pair(hyper["kernel_size"]), _pair(hyper["stride"]), _pair(hyper["dilation"])
    pad = hyper["padding"]
    if isinstance(pad, str):
        pad = tuple(_string_padding(k, pad, d)[0] for k, d in zip(ks, dl))
    pad = _pair(pad)
    B, C_, H, W = x.shape
    C_ //= hyper["groups"]
    OH = (H + 2 * pad[0] - dl[0] * (ks[0] - 1) - 1) // st[0] + 1
    OW = (W + 2 * pad[1] - dl[1] * (ks[1] - 1) - 1) // st[1] + 1
    Q, rows = C_ * ks[0] * ks[1], B * OH * OW
    if 4 * rows * Q > _PATCH_LIMIT_BYTES:
        return True
    if Q % 4 == 0:
        return False
    return not _hip.load().clo_gram_tall_supported(rows, Q, int(ones_col))  # tall-skinny: the streaming Gram kernel wins


# Conv2d input covariances of SMALL feature maps: pixel Gram + fold instead of a product over patches (csrc/conv.hip,
# `clo_patch_fold_f32`) wherever that is fewer flops -- (H W)^2 < OH OW (KH KW)^2 -- and the Gram fits `_PIXEL_GRAM_LIMIT`.
_PIXEL_GRAM = True
_PIXEL_GRAM_LIMIT_BYTES = 512 * 2**20


def _conv_geometry(hyper: dict, x: Tensor):
    ks, st, dl = _pair(hyper["kernel_size"]), _pair(hyper["stride"]), _pair(hyper["dilation"])
    pad = hyper["padding"]
    if isinstance(pad, str):
        pad = tuple(_string_padding(k, pad, d)[0] for k, d in zip(ks, dl))
    pad = _pair(pad)
    B, C_, H, W = x.shape
    C_ //= hyper["groups"]
    OH = (H + 2 * pad[0] - dl[0] * (ks[0] - 1) - 1) // st[0] + 1
    OW = (W + 2 * pad[1] - dl[1] * (ks[1] - 1) - 1) // st[1] + 1
    return ks, st, pad, dl, (B, C_, H, W), (OH, OW)


def _use_pixel_gram(hyper: dict, x: Tensor) -> bool:
    if not _PIXEL_GRAM or not _conv_patches_fusable(hyper):
        return False
    ks, st, pad, dl, (B, C_, H, W), (OH, OW) = _conv_geometry(hyper, x)
    if B == 0 or OH <= 0 or OW <= 0:
        return False
    hw, taps, pos = H * W, ks[0] * ks[1], OH * OW
    if hw * hw >= pos * taps * taps or 4 * (C_ * hw) ** 2 > _PIXEL_GRAM_LIMIT_BYTES:
        return False
    return bool(_hip.load().clo_patch_fold_supported(C_, H, W, ks[0], ks[1], OH, OW))


def _pixel_gram_accumulate(store: dict, key, x: Tensor, hyper: dict, n_data: int, ones_col: bool) -> None:
    """``store[key] += [P | 1]^T [P | 1] / (N_data * O1 O2)`` through the pixel Gram of ``x`` ([B, C, H, W])."""
    ks, st, pad, dl, (B, C_, H, W), (OH, OW) = _conv_geometry({**hyper, "groups": 1}, x)
    d = C_ * ks[0] * ks[1] + (1 if ones_col else 0)
    fresh = getattr(store, "fresh", None)
    Cm = store.get(key)
    first = Cm is None or (fresh is not None and key in fresh)
    if Cm is None:
        Cm = torch.empty(d, d, device=x.device, dtype=torch.float32)
        store[key] = Cm
    if fresh:
        fresh.discard(key)
    _hip.pixel_gram_accum(Cm, x, ks, st, pad, dl, alpha=1.0 / (n_data * OH * OW), beta=0.0 if first else 1.0,
                          ones_col=ones_col)


def _patch_gram_accumulate(store: dict, key, x: Tensor, hyper: dict, n_data: int, ones_col: bool) -> None:
    """``store[key] += [P | 1]^T [P | 1] / (N_data * O1 O2)`` for the patches ``P`` of ``x`` ([B, C, H, W])."""
    ks, st, dl = _pair(hyper["kernel_size"]), _pair(hyper["stride"]), _pair(hyper["dilation"])
    pad = hyper["padding"]
    if isinstance(pad, str):
        pad = tuple(_string_padding(k, pad, d)[0] for k, d in zip(ks, dl))
    pad = _pair(pad)
    B, C_, H, W = x.shape
    OH = (H + 2 * pad[0] - dl[0] * (ks[0] - 1) - 1) // st[0] + 1
    OW = (W + 2 * pad[1] - dl[1] * (ks[1] - 1) - 1) // st[1] + 1
    d = C_ * ks[0] * ks[1] + (1 if ones_col else 0)
    fresh = getattr(store, "fresh", None)
    Cm = store.get(key)
    first = Cm is None or (fresh is not None and key in fresh)
    if Cm is None:
        Cm = torch.empty(d, d, device=x.device, dtype=torch.float32)
        store[key] = Cm
    if fresh:
        fresh.discard(key)
    _hip.im2col_syrk_accum(Cm, x, ks, st, pad, dl, alpha=1.0 / (n_data * OH * OW), beta=0.0 if first else 1.0,
                           ones_col=ones_col)


# ------------------------------------------------------------------------------------------
# graph-captured factor build (the analogue of the reference's traced + compiled backend,
# computers/kfac_make_fx.py:26-111: trace the per-batch computation once, replay it per batch)
# ------------------------------------------------------------------------------------------
# One factor build of ResNet-18 is ~350 launches that the host dispatches in 7.5 ms (a bare gradient pass: 5.9 ms)
# while the kernels themselves need less.  A mini-batch of a given shape is therefore CAPTURED once as a hipGraph --
# forward pass, backpropagation, im2col / SYRK kernels on the factor stream (fork / join by events, so the overlap is
# part of the graph) -- and replayed for every later batch of that shape: the batch is copied into the graph's static
# input tensors, one graph launch recomputes everything from the LIVE parameters and buffers (their addresses are part
# of the cache key; their values are read at replay time), and the factors are copied / added out of the graph's
# static buffers.  Nothing is cached but the launch sequence.
_CAPTURE = True        # module knob (tools / tests flip it for A/B runs)
_CAPTURE_FORK = "coarse"   # "fine": one fork of the factor stream per hook, as in eager mode (see `_run_batch`)
_CAPTURE_G_CHUNK = 0   # coarse mode: gradient covariances per fork of the factor stream; 0 = inline on the main stream
#                        (measured, tools/probe_kfac_fork.py: inline 4.3 - 4.8 ms, chunks of 6: 4.9 - 5.8 ms, one fork at
#                        the end 4.75 ms per ResNet-18 build)
_CAPTURE_STREAM = None  # index of the package-wide worker stream to capture on (None: torch's own capture stream)
_CAPTURE_AFTER = 1     # eager runs of a configuration before it is captured
_CAPTURE_MAX = 4       # captured configurations kept (each holds its activations' memory pool + static factor buffers)
_CAPTURE_NOTES_MAX = 256   # bookkeeping entries (eager-run counters, "capture failed" marks) kept beside the graphs
# Branches of a captured build: a FIXED rule from the process configuration, nothing is timed (round 6; the round-5 code
# replayed candidate graphs against the wall clock).  2: the input covariances on a second branch beside the backward
# pass; 1: one in-order graph.  Measured on ResNet-18 / 512 rows inside the bench process (profiles/r06_kfac_capture_rule.txt):
# with the runtime's default of 4 hardware queues the two-branch graph replays in 4.2 - 4.5 ms (one branch 5.0), with 16
# queues in 6.5 - 6.8 ms (one branch 5.1: the branches' internal streams then sit on queues that share a dispatch pipe
# with the origin stream's).  So: two branches up to 4 queues, one beyond.
def _default_capture_branches() -> int:
    try:
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        queues = 4
    return 2 if queues <= 4 else 1


_CAPTURE_BRANCHES = int(os.environ.get("CLO_KFAC_CAPTURE_BRANCHES", "0")) or _default_capture_branches()
# Gradient covariances of a mini-batch in ONE grouped launch at the end of the backward pass (clo_syrk_grouped_f32) instead of
# one split-K product + reduction per layer: ResNet-18's 21 of them are 6 GFLOP that took 0.9 ms as 61 small launches.
_GROUP_G = os.environ.get("CLO_KFAC_GROUP_G", "1") == "1"
_CAPTURED: dict = {}   # signature -> int (eager runs so far) | _CapturedBatch | False (capture failed: stay eager)
_CAPTURE_GENERATORS: dict = {}
# type-2 / multi-sample MC builds (one batched backward pass under vmap, the callbacks fed afterwards) are captured too
# (round 6: with the capture done once and nothing replayed while capturing, their replays equal the eager build --
# test_captured_factor_build_equals_eager[type-2]; round 5 had excluded them)
_CAPTURE_MANUAL = os.environ.get("CLO_KFAC_CAPTURE_MANUAL", "1") == "1"
_CAPTURE_REPLAYS = 0   # graph launches so far (tests assert that a captured route really ran)


def _capture_generator(device: torch.device) -> torch.Generator:
    """ONE generator per device for every captured build: graphs register it (``register_generator_state``), so a
    replay draws from its current seed / offset and advances it exactly like the eager ops it recorded."""
    gen = _CAPTURE_GENERATORS.get(device)
    if gen is None:
        gen = _CAPTURE_GENERATORS[device] = torch.Generator(device=device)
    return gen


def reset_captured_builds() -> None:
    """Drop every captured factor build (frees their memory pools)."""
    _CAPTURED.clear()


def _note_capture_state(sig: tuple, value) -> None:
    """Bookkeeping entry (eager-run counter / False); the oldest ones go when there are too many -- a workload that
    keeps re-allocating its parameters would otherwise grow the table without bound."""
    _CAPTURED[sig] = value
    notes = [k for k, v in _CAPTURED.items() if not isinstance(v, _CapturedBatch)]
    for k in notes[: max(0, len(notes) - _CAPTURE_NOTES_MAX)]:
        del _CAPTURED[k]


class _CapturedBatch:
    """The KFAC factor computation of ONE mini-batch shape as a replayable hipGraph -- or as TWO (``split``): the forward
    pass with all input covariances, and the backward pass with the gradient covariances, so that a data-parallel build
    can start the all-reduce of the input covariances (98 % of the factor bytes) between them."""

    def __init__(self):
        self.graph = None        # whole batch, or the forward half of a split capture
        self.graph_bwd = None    # backward half of a split capture
        self.X = self.y = None
        self.store: _FactorStore | None = None
        self.model_ref = None
        self._keep = None

    @property
    def split(self) -> bool:
        return self.graph_bwd is not None

    @classmethod
    def _capture_once(cls, computer: "HipKFACComputer", X: Tensor, y: Tensor, mapping, sizes_a: dict, sizes_g: dict,
                      gen: torch.Generator, overlap: bool, split: bool = False):
        """One capture of the build (``overlap``: input covariances on the graph's second branch, else one branch;
        ``split``: two graphs, cut behind the last input covariance)."""
        import weakref

        global _OVERLAP
        self = cls()
        dev = computer.device
        self.X, self.y = X.clone(), y.clone()
        self.store = _FactorStore()
        self.store.preallocate({("a", k): d for k, d in sizes_a.items()} | {("g", k): d for k, d in sizes_g.items()},
                               dev, torch.float32)
        A, G = _FactorStore(), _FactorStore()
        for (which, k), view in self.store.items():
            (A if which == "a" else G)[k] = view
        A.fresh, G.fresh = set(sizes_a), set(sizes_g)   # first touch of a factor writes (beta = 0): no memset
        graphs = [torch.cuda.CUDAGraph()]
        graphs[0].register_generator_state(gen)
        state = gen.get_state()
        cap_stream = None if _CAPTURE_STREAM is None else side_stream(dev, _CAPTURE_STREAM)
        keep = _OVERLAP
        _OVERLAP = keep and overlap and not split
        # thread-local capture mode: only THIS thread's calls are checked against the capture.  Other threads of the process
        # -- the RCCL watchdog of a data-parallel job polls its events, worker threads allocate -- must neither fail nor
        # invalidate the capture (global mode, torch's default, does both).
        mode = "thread_local"
        ctx = [torch.cuda.graph(graphs[0], stream=cap_stream, capture_error_mode=mode)]
        entered = False

        def cut():
            """End of the forward half: every input covariance is complete (factors no hook wrote are zero)."""
            nonlocal entered
            for k in A.fresh:
                A[k].zero_()
            A.fresh = set()
            ctx[-1].__exit__(None, None, None)
            entered = False
            graphs.append(torch.cuda.CUDAGraph())
            graphs[-1].register_generator_state(gen)
            ctx.append(torch.cuda.graph(graphs[-1], pool=graphs[0].pool(), stream=cap_stream, capture_error_mode=mode))
            ctx[-1].__enter__()
            entered = True

        try:
            ctx[0].__enter__()
            entered = True
            with _use_params(computer._model_module, computer._params):
                # (a split build keeps the per-hook order of eager mode: the covariance of a layer's input is queued when
                # the layer runs, so the forward half ends with all of them done)
                computer._run_batch(self.X, self.y, mapping, A, G, after_forward=cut if split else None,
                                    coarse_fork=_CAPTURE_FORK == "coarse" and not split)
            for st in (A, G):            # factors no hook wrote (unused layers) are zero
                for k in st.fresh:
                    st[k].zero_()
            ctx[-1].__exit__(None, None, None)
            entered = False
        except BaseException as error:
            if entered:
                try:
                    ctx[-1].__exit__(type(error), error, None)
                except Exception:  # noqa: BLE001 - the capture is already lost; report the first error
                    pass
            raise
        finally:
            _OVERLAP = keep
            gen.set_state(state)             # (capture advanced the generator without drawing anything)
        self.graph = graphs[0]
        self.graph_bwd = graphs[1] if split else None
        self.n_a = self.store.end_of(("a", k) for k in sizes_a)
        self.model_ref = weakref.ref(computer._model_module)
        return self

    @classmethod
    def capture(cls, computer: "HipKFACComputer", X: Tensor, y: Tensor, mapping, sizes_a: dict, sizes_g: dict,
                gen: torch.Generator, sig: tuple, split: bool = False):
        """Capture ONCE, by the fixed rule `_CAPTURE_BRANCHES` (no candidate graphs, no timing replays: a capture records
        launches and executes nothing, so module buffers and RNG streams see every mini-batch exactly once, as in eager
        mode).  On any failure the configuration is marked uncapturable (False) and the caller runs it eagerly."""
        try:
            self = cls._capture_once(computer, X, y, mapping, sizes_a, sizes_g, gen, _CAPTURE_BRANCHES >= 2, split)
        except Exception as error:  # noqa: BLE001 - any capture problem means: stay on the eager route
            from warnings import warn

            warn(f"KFAC factor build: hipGraph capture failed ({type(error).__name__}: {error}); this configuration "
                 "keeps the eager route.", stacklevel=3)
            _note_capture_state(sig, False)
            torch.cuda.synchronize(computer.device)
            return False
        # bounded cache; ids of dead models must not alias new ones
        for k in [k for k, v in _CAPTURED.items() if isinstance(v, _CapturedBatch) and v.model_ref() is None]:
            del _CAPTURED[k]
        live = [k for k, v in _CAPTURED.items() if isinstance(v, _CapturedBatch)]
        for k in live[: max(0, len(live) + 1 - _CAPTURE_MAX)]:
            del _CAPTURED[k]
        _CAPTURED[sig] = self
        return self

    def replay(self, X: Tensor, y: Tensor) -> None:
        self.replay_forward(X, y)
        self.replay_backward()

    def replay_forward(self, X: Tensor, y: Tensor) -> None:
        """Whole batch, or (split capture) the forward pass + every input covariance."""
        global _CAPTURE_REPLAYS
        self.X.copy_(X)
        self.y.copy_(y)
        self.graph.replay()
        _CAPTURE_REPLAYS += 1

    def replay_backward(self) -> None:
        global _CAPTURE_REPLAYS
        if self.graph_bwd is not None:
            self.graph_bwd.replay()
            _CAPTURE_REPLAYS += 1


class HipKFACComputer(EmpiricalRiskMixin):
    """Computes KFAC's Kronecker factors ``A_l`` (input covariance) and ``G_l`` (output-gradient
    covariance) for every Linear / Conv2d parameter group."""

    _SUPPORTED_LOSSES = (MSELoss, CrossEntropyLoss, BCEWithLogitsLoss)
    _SUPPORTED_MODULES = (Linear, Conv2d)
    _SUPPORTED_FISHER_TYPE = FisherType
    _SUPPORTED_KFAC_APPROX = KFACType
    NEEDS_NUM_PER_EXAMPLE_LOSS_TERMS: bool = True
    _REQUIRES_MODULE: bool = True

    def __init__(
        self,
        model_func: Module,
        loss_func: MSELoss | CrossEntropyLoss | BCEWithLogitsLoss,
        params: dict[str, Tensor],
        data: Iterable[tuple[Tensor | MutableMapping, Tensor]],
        progressbar: bool = False,
        check_deterministic: bool = True,
        seed: int = 2_147_483_647,
        fisher_type: str = FisherType.MC,
        mc_samples: int = 1,
        kfac_approx: str = KFACType.EXPAND,
        num_per_example_loss_terms: int | None = None,
        separate_weight_and_bias: bool = True,
        num_data: int | None = None,
        batch_size_fn: Callable[[MutableMapping | Tensor], int] | None = None,
        distributed: bool = False,
    ):
        """``distributed=True``: ``data`` is this rank's shard, ``num_data`` the GLOBAL count; the
        factors (and EKFAC's corrected eigenvalues) are summed over ranks with one packed
        all-reduce each (``curvlinops_amd.dist``).  All other arguments as in the reference."""
        if self._REQUIRES_MODULE and not isinstance(model_func, Module):
            raise ValueError(
                "The hooks-based backends require model_func to be an nn.Module."
            )
        if distributed and num_data is None:
            raise ValueError("distributed=True needs the global num_data.")
        self._distributed = distributed
        if not isinstance(loss_func, self._SUPPORTED_LOSSES):
            raise ValueError(f"Invalid loss: {loss_func}. Supported: {self._SUPPORTED_LOSSES}.")
        if fisher_type not in self._SUPPORTED_FISHER_TYPE:
            raise ValueError(f"Invalid fisher_type: {fisher_type}. Supported: {self._SUPPORTED_FISHER_TYPE}.")
        if fisher_type != FisherType.MC and mc_samples != 1:
            raise ValueError(
                f"Invalid mc_samples: {mc_samples}. Only mc_samples=1 is supported for "
                "`fisher_type != FisherType.MC`."
            )
        if kfac_approx not in self._SUPPORTED_KFAC_APPROX:
            raise ValueError(f"Invalid kfac_approx: {kfac_approx}. Supported: {self._SUPPORTED_KFAC_APPROX}.")
        self._seed = seed
        self._generator: torch.Generator | None = None
        self._separate_weight_and_bias = separate_weight_and_bias
        self._fisher_type = fisher_type
        self._mc_samples = mc_samples
        self._kfac_approx = kfac_approx
        randomness = "different" if fisher_type == FisherType.MC else "same"
        self._grad_outputs_computer = vmap(
            make_grad_output_fn(loss_func, fisher_type, mc_samples), in_dims=(0, 0, None), out_dims=1,
            randomness=randomness,
        )
        super().__init__(
            model_func, loss_func, params, data, progressbar=progressbar,
            check_deterministic=check_deterministic, num_data=num_data,
            num_per_example_loss_terms=num_per_example_loss_terms, batch_size_fn=batch_size_fn,
        )

    # ------------------------------------------------------------------ public
    def compute(self):
        captured = self._compute_captured()
        if captured is not None:
            return captured
        with _use_params(self._model_module, self._params):
            return self._compute_kronecker_factors()

    @classmethod
    def compute_parameter_groups(cls, params: dict[str, Tensor], model: Module,
                                 separate_weight_and_bias: bool = True) -> list[ParamGroup]:
        """One group per supported layer (joint W+b) or per parameter (separate)."""
        role = {"weight": "W", "bias": "b"}
        wanted = set(params.keys())
        groups: list[ParamGroup] = []
        seen: set[str] = set()
        for mod_name, mod in model.named_modules():
            if not isinstance(mod, cls._SUPPORTED_MODULES):
                continue
            roles: ParamGroup = {}
            for p_name, _ in mod.named_parameters(recurse=False):
                full = f"{mod_name}.{p_name}" if mod_name else p_name
                if full in wanted:
                    roles[role[p_name]] = full
                    seen.add(full)
            if roles:
                groups.extend([{r: n} for r, n in roles.items()] if separate_weight_and_bias else [roles])
        if wanted - seen:
            raise NotImplementedError(
                f"Parameters {wanted - seen} are not in supported layers ({cls._SUPPORTED_MODULES})."
            )
        return groups

    # ------------------------------------------------------------------ internals
    def _module_of(self, group: ParamGroup) -> Module:
        name = next(iter(group.values()))
        return self._model_module.get_submodule(name.rsplit(".", 1)[0] if "." in name else "")

    def _rearrange_output(self, output: Tensor, y: Tensor) -> tuple[Tensor, Tensor]:
        return flatten_output_and_labels(output, y, self._loss_func)

    def _compute_kronecker_factors(self):
        mapping = self.compute_parameter_groups(self._params, self._model_module, self._separate_weight_and_bias)
        A: dict[ParamGroupKey, Tensor] = _FactorStore()
        G: dict[ParamGroupKey, Tensor] = _FactorStore()
        if self._distributed:
            # all factors of this rank in ONE flat buffer, accumulated in place and all-reduced in
            # place (A_l first, then G_l): sizes follow from the layer shapes
            sizes_a, sizes_g = self._factor_sizes(mapping)
            both = _FactorStore()
            both.preallocate({("a", k): d for k, d in sizes_a.items()} | {("g", k): d for k, d in sizes_g.items()},
                             self.device, self.dtype)
            for (which, k), view in both.items():
                (A if which == "a" else G)[k] = view
            A.fresh, G.fresh = set(sizes_a), set(sizes_g)
            flat = both.flat
            n_a = both.end_of(("a", k) for k in sizes_a)  # flat[:n_a] = all A_l, flat[n_a:] = all G_l
        self._generator = seed_generator(self._generator, self.device, self._seed)
        work_a = None
        batches = iter(self._loop_over_data(desc="KFAC matrices"))
        nxt = next(batches, None)
        while nxt is not None:
            (X, y), nxt = nxt, next(batches, None)
            after_forward = None
            if nxt is None and self._distributed:
                # The input covariances are complete once the LAST forward pass has run: their
                # all-reduce (all but ~2 % of the factor bytes: ResNet-18 369 of 376 MB) starts now and
                # travels over xGMI while this rank is still backpropagating.
                def after_forward():
                    nonlocal work_a
                    work_a = self._start_input_factor_allreduce(A, flat, n_a)
            self._run_batch(X, y, mapping, A, G, after_forward)
        if self._distributed:
            from curvlinops_amd.dist import allreduce_flat_

            if work_a is None:  # no batch on this rank: its input covariances are zero
                work_a = self._start_input_factor_allreduce(A, flat, n_a)
            for k in G.fresh:   # a factor no batch contributed to (empty shard) is zero
                G[k].zero_()
            G.fresh = set()
            allreduce_flat_(flat[n_a:])
            if work_a is not True:
                work_a.wait()
        if self._fisher_type == FisherType.FORWARD_ONLY:
            for group in mapping:
                p = self._params[next(iter(group.values()))]
                G[tuple(group.values())] = torch.eye(p.shape[0], dtype=p.dtype, device=self.device)
        return dict(A), dict(G), mapping

    def _run_batch(self, X, y: Tensor, mapping, A, G, after_forward=None, coarse_fork: bool = False) -> None:
        """One mini-batch: forward pass with the input hooks (``A`` accumulation), backpropagation of the Fisher type's
        vectors with the output-gradient hooks (``G`` accumulation); the factor stream is joined at the end.

        ``coarse_fork`` (graph capture): instead of one fork per hook -- 42 cross-stream edges in a ResNet-18 graph, each
        a barrier packet with a completion signal in the main queue when the two branches sit on different hardware
        queues -- the input-covariance work of ALL layers is queued on the factor stream behind ONE event at the end of
        the forward pass (the layer inputs are alive until the backward pass anyway) and runs beside the backward pass;
        the (small) gradient covariances run inline on the main stream (or in chunks of `_CAPTURE_G_CHUNK` layers, one
        fork per chunk)."""
        self._deferred_inputs = [] if coarse_fork else None
        self._inline_grads = coarse_fork
        self._deferred_grads = []
        self._grouped_g = []
        handles = []
        for group in mapping:
            mod = self._module_of(group)
            hyper = _conv_hyperparams(mod)
            if "W" in group:
                handles.append(mod.register_forward_pre_hook(partial(self._input_hook, group=group, hyper=hyper, store=A)))
            handles.append(mod.register_forward_hook(partial(self._output_hook, group=group, hyper=hyper, store=G)))
        self._hooked_outputs = []
        try:
            output = self._model_module(X)
            if self._deferred_inputs:
                jobs, self._deferred_inputs = self._deferred_inputs, None
                with _factor_stream(output):
                    for job in jobs:
                        job()
                del jobs
            if after_forward is not None:
                after_forward()
            output, y = self._rearrange_output(output, y)
            self._backpropagate(output, y)
            self._flush_deferred_grads(None)
            self._flush_grouped_grads()
        finally:
            for h in handles:
                h.remove()
            self._hooked_outputs = []
            self._deferred_inputs = None
            self._deferred_grads = []
            self._grouped_g = None     # (outside a batch `_grad_job` accumulates at once)
            self._inline_grads = False
            _join_factor_stream(self.device)

    def _factor_sizes(self, mapping) -> tuple[dict, dict]:
        """``{group key: order}`` of the input / gradient covariances this computer builds."""
        sizes_a, sizes_g = {}, {}
        for group in mapping:
            mod, key = self._module_of(group), tuple(group.values())
            if "W" in group:
                sizes_a[key] = mod.weight[0].numel() + (1 if "b" in group else 0)
            if self._fisher_type != FisherType.FORWARD_ONLY:
                sizes_g[key] = self._params[next(iter(group.values()))].shape[0]
        return sizes_a, sizes_g

    # ------------------------------------------------------------------ graph-captured build
    def _loss_signature(self) -> tuple:
        """The loss configuration a captured graph bakes in (class weights, label smoothing, ignore_index, pos_weight ...):
        plain attributes by value, tensors by address."""
        loss = self._loss_func
        items = []
        for name, val in sorted({**vars(loss), **loss._buffers}.items()):
            if name.startswith("_") or name == "training":
                continue
            items.append((name, (val.data_ptr(), tuple(val.shape), val.dtype) if isinstance(val, Tensor) else repr(val)))
        return (type(loss).__module__, type(loss).__qualname__, tuple(items))

    def _capture_signature(self, X: Tensor, y: Tensor) -> tuple | None:
        """Everything a captured batch bakes in: shapes, normalisation constants, the Fisher type, the loss and its
        settings, and the ADDRESSES of every tensor its kernels read in place (parameters, buffers) -- values stay live,
        a swapped storage re-captures."""
        model = self._model_module
        if not (isinstance(X, Tensor) and is_native_tensor(X) and isinstance(y, Tensor) and y.is_cuda
                and X.device == self.device and y.device == self.device and X.shape[0] > 0):
            return None
        tensors = list(self._params.values())
        mods = list(model.modules())
        for m in mods:
            tensors.extend(m._parameters.values())
            tensors.extend(m._buffers.values())
        if any(t is not None and not (t.is_cuda and t.device == self.device) for t in tensors):
            return None
        return (
            id(model), type(self).__name__, self._loss_signature(),
            str(self._fisher_type), self._mc_samples, str(self._kfac_approx), self._separate_weight_and_bias,
            self._N_data, self._num_per_example_loss_terms, tuple(self._params.keys()),
            tuple(X.shape), tuple(y.shape), y.dtype, str(self.device), _FUSED_IM2COL, _FAST_BN, _OVERLAP, _PIXEL_GRAM, _CAPTURE_FORK, _CAPTURE_G_CHUNK,
            _CAPTURE_BRANCHES, bool(self._distributed), _GROUP_G,
            tuple(m.training for m in mods),
            tuple((0, 0) if t is None else (t.data_ptr(), t.dtype) for t in tensors),
        )

    def _compute_captured(self):
        """The factor build with every mini-batch replayed as a hipGraph (``_CapturedBatch``) where one is available or
        can be captured; None: take the eager route (not eligible, or capture failed once for this configuration).

        ``distributed=True`` runs the SAME graphs (round 6; the reference has one code path for any data list,
        ``kfac_hooks.py:219-224``): the shard's batches are replayed, the LAST one as a split capture -- forward pass +
        input covariances, then the asynchronous all-reduce of the input-covariance part of the flat buffer is started,
        and the backward half replays while it travels; the gradient covariances are reduced behind it.  Collectives are
        issued in the same order (A, then G) as on the eager route, so ranks may mix routes."""
        # (several backpropagated vectors per datum -- type-2, multi-sample MC -- run ONE batched backward pass under
        # vmap and feed the callbacks afterwards; `_CAPTURE_MANUAL` gates their capture)
        if not (_CAPTURE and self._CAPTURABLE and (not self._manual_callbacks or _CAPTURE_MANUAL) and not self._progressbar
                and isinstance(self._data, (list, tuple)) and self._data and self.device.type == "cuda"
                and self.dtype == torch.float32 and isinstance(self._model_module, Module)):
            return None
        dist_on = bool(self._distributed)
        sigs = []
        for X, y in self._data:
            sig = self._capture_signature(X, y)
            if sig is None or _CAPTURED.get(sig, 0) is False:
                return None
            sigs.append(sig)
        mapping = self.compute_parameter_groups(self._params, self._model_module, self._separate_weight_and_bias)
        sizes_a, sizes_g = self._factor_sizes(mapping)
        gen = _capture_generator(self.device)
        gen.manual_seed(self._seed)
        self._generator = gen
        out = _FactorStore()
        out.preallocate({("a", k): d for k, d in sizes_a.items()} | {("g", k): d for k, d in sizes_g.items()},
                        self.device, self.dtype)
        A, G = _FactorStore(), _FactorStore()
        for (which, k), view in out.items():
            (A if which == "a" else G)[k] = view
        A.fresh, G.fresh = set(sizes_a), set(sizes_g)
        n_a = out.end_of(("a", k) for k in sizes_a)   # out.flat[:n_a] = all A_l, out.flat[n_a:] = all G_l
        work_a = None

        def take(entry, lo: int, hi: int, stores) -> None:
            """out.flat[lo:hi] (+)= the graph's static factors; `stores`: the stores whose views lie in that range."""
            if all(len(st.fresh) == len(st) for st in stores):
                out.flat[lo:hi].copy_(entry.store.flat[lo:hi])
            else:
                for st in stores:
                    for k in st.fresh:
                        st[k].zero_()
                out.flat[lo:hi].add_(entry.store.flat[lo:hi])
            for st in stores:
                st.fresh = set()

        last = len(self._data) - 1
        for i, ((X, y), sig) in enumerate(zip(self._data, sigs)):
            entry = _CAPTURED.get(sig, 0)
            if isinstance(entry, _CapturedBatch) and entry.model_ref() is not self._model_module:
                entry = 0   # the id of a dead model, recycled
            if isinstance(entry, int):
                # a configuration is run eagerly the first time it is seen (library warm-up: MIOpen's solver search,
                # workspaces) and captured when it comes back
                if entry >= _CAPTURE_AFTER:
                    entry = _CapturedBatch.capture(self, X, y, mapping, sizes_a, sizes_g, gen, sig, split=dist_on)
                else:
                    _note_capture_state(sig, entry + 1)
            overlap_now = dist_on and i == last   # the shard's input covariances are complete after THIS forward pass
            if isinstance(entry, _CapturedBatch):
                if overlap_now and entry.split:
                    entry.replay_forward(X, y)
                    take(entry, 0, n_a, (A,))
                    work_a = self._start_input_factor_allreduce(A, out.flat, n_a)
                    entry.replay_backward()
                    take(entry, n_a, out.flat.numel(), (G,))
                else:
                    entry.replay(X, y)
                    take(entry, 0, out.flat.numel(), (A, G))
            else:
                after_forward = None
                if overlap_now:
                    def after_forward():
                        nonlocal work_a
                        work_a = self._start_input_factor_allreduce(A, out.flat, n_a)
                with _use_params(self._model_module, self._params):
                    self._run_batch(X, y, mapping, A, G, after_forward)
        for st in (A, G):
            for k in st.fresh:
                st[k].zero_()
            st.fresh = set()
        if dist_on:
            from curvlinops_amd.dist import allreduce_flat_

            if work_a is None:
                work_a = self._start_input_factor_allreduce(A, out.flat, n_a)
            allreduce_flat_(out.flat[n_a:])
            if work_a is not True:
                work_a.wait()
        if self._fisher_type == FisherType.FORWARD_ONLY:
            for group in mapping:
                p = self._params[next(iter(group.values()))]
                G[tuple(group.values())] = torch.eye(p.shape[0], dtype=p.dtype, device=self.device)
        return dict(A), dict(G), mapping

    _CAPTURABLE = True

    def _start_input_factor_allreduce(self, A, flat: Tensor, n_a: int):
        """Asynchronous in-place all-reduce of the input-covariance part of the flat factor buffer, ordered
        after the SYRKs that produced it (they run on the factor stream): returns the work handle (True if
        there is nothing to wait for)."""
        import torch.distributed as tdist

        from curvlinops_amd.dist import is_distributed

        part = flat[:n_a]
        with _factor_stream(part):
            for k in A.fresh:
                A[k].zero_()
            A.fresh = set()
            if not is_distributed() or n_a == 0:
                return True
            return tdist.all_reduce(part, op=tdist.ReduceOp.SUM, async_op=True)

    def _backpropagate(self, output: Tensor, y: Tensor) -> None:
        """Backpropagate V vectors per datum (0 forward-only, 1 empirical, M for MC, C for
        type-2); the tensor hooks registered on the layer outputs see each of them."""
        if output.ndim != 2 or y.ndim not in {1, 2}:
            raise ValueError(f"Only 2d output and 1d/2d target are supported. Got {output.ndim=} and {y.ndim=}.")
        grad_outputs = self._grad_outputs_computer(output.detach(), y, self._generator)
        if self._loss_func.reduction == "mean":
            grad_outputs.mul_(1.0 / output.shape[0])
        # Differentiate w.r.t. the hooked layer outputs themselves: the tensor hooks see exactly the
        # output-gradients KFAC needs and autograd never launches a weight-gradient kernel (the
        # reference differentiates w.r.t. the parameters, kfac_hooks.py:236-289, and discards them).
        hooked = [(o, cb) for o, cb in self._hooked_outputs if o.requires_grad]
        self._hooked_outputs = []
        wrt = [o for o, _ in hooked]
        if not wrt:
            module_params = dict(self._model_module.named_parameters())
            wrt = [module_params[n] for n in self._params]
        V = grad_outputs.shape[0]
        if self._manual_callbacks and hooked:
            # V > 1 vectors per datum (type-2, several MC samples): ONE batched backward pass instead
            # of V sequential ones; the callbacks then see the V slices of every layer's gradient
            try:
                grads = torch.autograd.grad(output, wrt, grad_outputs=grad_outputs, is_grads_batched=True,
                                            allow_unused=True, retain_graph=True)
                for (_, cb), g in zip(hooked, grads):
                    if g is not None:
                        cb(g, stacked=True)  # [V, B, ...]: one accumulation for all V vectors
                return
            except RuntimeError:
                pass  # an op without a batching rule: fall back to V sequential passes
            for v in range(V):
                grads = torch.autograd.grad(output, wrt, grad_outputs=grad_outputs[v], retain_graph=v < V - 1,
                                            allow_unused=True)
                for (_, cb), g in zip(hooked, grads):
                    if g is not None:
                        cb(g)
            return
        for v in range(V):
            torch.autograd.grad(output, wrt, grad_outputs=grad_outputs[v], retain_graph=v < V - 1,
                                allow_unused=True)

    @property
    def _manual_callbacks(self) -> bool:
        """More than one backpropagated vector per datum: gradients are taken by one batched
        ``autograd.grad`` and handed to the callbacks afterwards (tensor hooks would fire inside
        ``vmap`` with batched tensors); with a single vector the tensor hooks fire DURING the
        backward pass, which lets the factor stream overlap it."""
        if self._fisher_type == FisherType.MC:
            return self._mc_samples > 1
        return self._fisher_type == FisherType.TYPE2

    def _track_output(self, output: Tensor, callback) -> None:
        if not self._manual_callbacks:
            output.register_hook(callback)
        self._hooked_outputs.append((output, callback))

    def _input_hook(self, module, inputs, group, hyper, store) -> None:
        if len(inputs) != 1:
            raise ValueError("Modules with multiple inputs are not supported.")
        if getattr(self, "_deferred_inputs", None) is not None and is_native_tensor(inputs[0]):
            x_keep = inputs[0].data.detach()
            side = _FACTOR_STREAMS.get(x_keep.device) or side_stream(x_keep.device, 0)
            x_keep.record_stream(side)
            self._deferred_inputs.append(partial(self._input_job, x_keep, group, hyper, store))
            return
        with _factor_stream(inputs[0]):
            self._input_job(inputs[0].data.detach(), group, hyper, store)

    def _input_job(self, x_in: Tensor, group, hyper, store) -> None:
        joint = "W" in group and "b" in group
        if (hyper and self._kfac_approx == KFACType.EXPAND and is_native_tensor(x_in) and x_in.dim() == 4
                and _use_pixel_gram(hyper, x_in)):
            # Conv2d on a small feature map: A from the pixel Gram X^T X (X = x as [B, C H W]) folded over the taps --
            # fewer flops than the product over patches, no patch matrix (csrc/conv.hip)
            _pixel_gram_accumulate(store, tuple(group.values()), _group_mean(x_in, hyper["groups"]), hyper,
                                   self._N_data, ones_col=joint)
            return
        if (hyper and self._kfac_approx == KFACType.EXPAND and is_native_tensor(x_in) and x_in.dim() == 4
                and _use_fused_patches(hyper, x_in, joint)):
            # Conv2d, KFAC-expand: the patch matrix [B O1 O2, C K1 K2] is generated inside the SYRK's
            # tile loader and never written (reference materialises it, kfac_utils.py:78-121)
            _patch_gram_accumulate(store, tuple(group.values()), _group_mean(x_in, hyper["groups"]), hyper,
                                   self._N_data, ones_col=joint)
            return
        x = input_to_weight_sharing_format(x_in, self._kfac_approx, hyper)
        shared = x.shape[1]
        _gram_accumulate(store, tuple(group.values()), x.reshape(-1, x.shape[-1]),
                         1.0 / (self._N_data * shared), ones_col=joint)

    def _output_hook(self, module, inputs, output, group, hyper, store) -> None:
        self._track_output(output, partial(self._grad_hook, group=group, hyper=hyper, store=store))

    def _grad_hook(self, grad_output: Tensor, group, hyper, store, stacked: bool = False) -> None:
        g = grad_output.data.detach()
        batch_size = g.shape[1] if stacked else g.shape[0]
        if stacked:  # [V, B, ...] -> [V B, ...]: the V vectors are just more rows of the Gram matrix
            g = g.flatten(0, 1)
        corr = compute_loss_correction(batch_size, self._num_per_example_loss_terms,
                                       self._loss_func.reduction, self._N_data)
        if self._group_this_grad(g, hyper):
            # kept until the backward pass is over: ONE launch then forms every layer's G_l (`_flush_grouped_grads`).
            # Formatted HERE, on the stream autograd produced the gradient on -- the flush runs on that stream too (a
            # copy made under `_factor_stream` would race with it).
            g2d = grad_to_weight_sharing_format(g, self._kfac_approx, hyper)
            g2d = g2d.reshape(-1, g2d.shape[-1])
            key = tuple(group.values())
            if any(k == key and st is store for st, k, _, _ in self._grouped_g):
                self._flush_grouped_grads()   # a second vector for the same factor: the first group goes first
            self._grouped_g.append((store, key, g2d if g2d.stride(-1) == 1 else g2d.contiguous(), corr))
            return
        if getattr(self, "_inline_grads", False) and is_native_tensor(g):
            # coarse fork (graph capture): the (small) gradient covariances run inline on the main stream
            # (`_CAPTURE_G_CHUNK` = 0), or those of `_CAPTURE_G_CHUNK` consecutive layers share ONE fork of the factor
            # stream (the gradients are kept alive until their chunk has been queued)
            if _CAPTURE_G_CHUNK <= 0:
                self._grad_job(g, corr, group, hyper, store)
                return
            side = _FACTOR_STREAMS.get(g.device) or side_stream(g.device, 0)
            g.record_stream(side)
            self._deferred_grads.append(partial(self._grad_job, g, corr, group, hyper, store))
            if len(self._deferred_grads) >= _CAPTURE_G_CHUNK:
                self._flush_deferred_grads(g)
            return
        with _factor_stream(g):
            self._grad_job(g, corr, group, hyper, store)

    _GROUP_MIN_D = 48   # narrower factors (LeNet's 6 / 16 channels, a 10-class head) stay on the tall-skinny Gram kernel:
    #                     a 64-wide MFMA tile would be mostly padding (LeNet-5 type-2 build 2.8 -> 9.5 ms when they were grouped)

    def _group_this_grad(self, g: Tensor, hyper) -> bool:
        if not (_GROUP_G and getattr(self, "_grouped_g", None) is not None and is_native_tensor(g)):
            return False
        d_out = g.shape[1] if hyper else g.shape[-1]   # Conv2d: (rows, C, H, W); Linear: (rows, [S,] d)
        return d_out >= self._GROUP_MIN_D and _hip.has("clo_syrk_grouped_f32")

    def _grad_job(self, g: Tensor, corr: float, group, hyper, store) -> None:
        g = grad_to_weight_sharing_format(g, self._kfac_approx, hyper)
        _gram_accumulate(store, tuple(group.values()), g.reshape(-1, g.shape[-1]), corr, ones_col=False)

    def _flush_grouped_grads(self) -> None:
        """``G_l (+)= corr g_l^T g_l`` for every pending layer in one grouped launch (on the stream the gradients were
        produced on; a factor's first contribution writes, later ones accumulate)."""
        pending, self._grouped_g = getattr(self, "_grouped_g", None) or [], []
        if not pending:
            return
        Cs, Xs, alphas, betas = [], [], [], []
        for store, key, g2d, corr in pending:
            d = g2d.shape[1]
            fresh = getattr(store, "fresh", None)
            C = store.get(key)
            first = C is None or (fresh is not None and key in fresh)
            if C is None:
                C = store[key] = torch.empty(d, d, device=g2d.device, dtype=torch.float32)
            if fresh:
                fresh.discard(key)
            Cs.append(C), Xs.append(g2d), alphas.append(corr), betas.append(0.0 if first else 1.0)
        _hip.syrk_grouped(Cs, Xs, alphas, betas)

    def _flush_deferred_grads(self, ready: Tensor | None) -> None:
        jobs, self._deferred_grads = getattr(self, "_deferred_grads", []), []
        if not jobs:
            return
        if ready is None:   # after the backward pass: everything on the current stream is ready
            ready = torch.empty(0, device=self.device, dtype=torch.float32)
        with _factor_stream(ready):
            for job in jobs:
                job()


# ------------------------------------------------------------------------------------------
# EKFAC
# ------------------------------------------------------------------------------------------
def compute_eigenvalue_correction(g: Tensor, Qg: Tensor, a: Tensor | None, Qa: Tensor | None) -> Tensor:
    """``sum_{v,n} (Q_g^T (sum_s g_vns a_ns^T) Q_a)^2`` (``[d_out, d_in]``), or for a bias-only
    group ``sum_{v,n} (Q_g^T sum_s g_vns)^2`` (``[d_out]``).

    ``g``: ``[V, B, S, d_out]``; ``a``: ``[B, S, d_in]`` (reference
    ``computers/ekfac_hooks.py:25-238`` -- both of its strategies compute this quantity)."""
    identity = Qg is None  # no rotation at all: squared per-example gradients (the GGN diagonal)
    if not identity and (a is None) != (Qa is None):
        raise ValueError(f"Both (a, aaT_eigvecs) must be None or Tensor. Got {(type(a), type(Qa))}.")
    V, B, S, d1 = g.shape
    native = is_native_tensor(g) and (identity or is_native_tensor(Qg))
    if a is None:
        gs = g.sum(dim=2).reshape(V * B, d1)
        rot = gs if identity else (_hip.gemm(gs.contiguous(), Qg) if native else gs @ Qg)
        return rot.square().sum(dim=0)
    d2 = a.shape[-1]
    if native and not identity and is_native_tensor(a) and is_native_tensor(Qa) and V * B * S < 2**31:
        # both rotations and the squared-product reduction in ONE foreign call
        from curvlinops_amd.kronecker import _contiguous_pair

        fc = _contiguous_pair([Qg, Qa], False)
        if fc is not None:
            return _hip.ekfac_correction(g.contiguous(), fc[0], a.contiguous(), fc[1], rows=fc[2])
    if native:
        if identity:
            g_rot, a_rot = g.contiguous(), a.contiguous()
        else:
            g_rot = _hip.gemm(g.reshape(V * B * S, d1).contiguous(), Qg).view(V, B, S, d1)
            a_rot = _hip.gemm(a.reshape(B * S, d2).contiguous(), Qa).view(B, S, d2)
        if S == 1:
            # no weight sharing: the per-example gradient is the outer product g_n a_n^T, its square the
            # outer product of the squares -- sum_{v,n} = (sum_v g_vn^2)^T (a_n^2), ONE GEMM with K = B
            # instead of B rank-1 products (ResNet-18 layer4 / fc, every Linear layer of an MLP)
            g2 = g_rot.view(V, B, d1).square().sum(dim=0)
            return _hip.gemm(g2.T, a_rot.view(B, d2).square())
        out = torch.zeros(d1, d2, device=g.device, dtype=torch.float32)
        for v in range(V):
            # per-example products P_n = g_rot_n^T a_rot_n, squared and summed over n, fused
            _hip.gemm_sqsum(g_rot[v].transpose(1, 2), a_rot, out, beta=1.0)
        return out
    g_rot = g if identity else g @ Qg
    a_rot = a if identity else a @ Qa
    if S == 1:
        return (g_rot.reshape(V, B, d1).square().sum(dim=0)).T @ a_rot.reshape(B, d2).square()
    per_example = torch.einsum("vnsi,nsj->vnij", g_rot, a_rot)
    return per_example.square_().sum(dim=(0, 1))


class HipEKFACComputer(HipKFACComputer):
    """KFAC factors -> their eigenbases -> eigenvalues re-fitted in that basis (second sweep)."""

    _SUPPORTED_FISHER_TYPE = (FisherType.TYPE2, FisherType.MC, FisherType.EMPIRICAL)

    def _rearrange_output(self, output: Tensor, y: Tensor) -> tuple[Tensor, Tensor]:
        if output.ndim != 2 or y.ndim not in {1, 2}:
            raise ValueError(
                f"Only 2d output and 1d/2d target are supported for EKFAC. Got {output.ndim=} and {y.ndim=}."
            )
        return output, y

    def compute(self):
        factors = self._compute_captured()   # (the factor sweep as a replayed hipGraph where one is available, 3.6)
        with _use_params(self._model_module, self._params):
            A, G, mapping = factors if factors is not None else self._compute_kronecker_factors()
            keys = [("a", k) for k in A] + [("g", k) for k in G]
            mats = [A[k] if w == "a" else G[k] for w, k in keys]
            if self._distributed:
                # identical factors on every rank: shard the eigendecompositions by factor
                from curvlinops_amd.dist import sharded_factor_map

                bases = sharded_factor_map(mats, linalg_native.eigh_many, lambda n: [(n,), (n, n)])
            else:
                bases = linalg_native.eigh_many(mats)
            Qa = {k: q[1] for (w, k), q in zip(keys, bases) if w == "a"}
            Qg = {k: q[1] for (w, k), q in zip(keys, bases) if w == "g"}
            lam = self._eigenvalue_correction(Qa, Qg, mapping)
        return Qa, Qg, lam, mapping

    def _eigenvalue_correction(self, Qa, Qg, mapping):
        lam: dict[ParamGroupKey, Tensor] = {}
        handles = []
        for group in mapping:
            mod = self._module_of(group)
            handles.append(mod.register_forward_hook(partial(self._corr_output_hook, group=group, Qa=Qa, Qg=Qg, lam=lam)))
        self._generator = seed_generator(self._generator, self.device, self._seed)
        self._hooked_outputs = []
        try:
            for X, y in self._loop_over_data(desc="Eigenvalue correction"):
                output = self._model_module(X)
                output, y = self._rearrange_output(output, y)
                self._backpropagate(output, y)
        finally:
            for h in handles:
                h.remove()
        if self._distributed:
            from curvlinops_amd.dist import allreduce_tensors_

            allreduce_tensors_(list(lam.values()))
        return lam

    def _corr_output_hook(self, module, inputs, output, group, Qa, Qg, lam) -> None:
        self._track_output(output, partial(self._corr_grad_hook, module=module, inputs=inputs, group=group,
                                           Qa=Qa, Qg=Qg, lam=lam))

    def _corr_grad_hook(self, grad_output: Tensor, module, inputs, group, Qa, Qg, lam, stacked: bool = False) -> None:
        if len(inputs) != 1:
            raise ValueError("Modules with multiple inputs are not supported.")
        g = grad_output.data.detach()
        hyper = _conv_hyperparams(module)
        if stacked:  # [V, B, ...]: format every vector, keep the leading V axis
            batch_size = g.shape[1]
            g = torch.stack([grad_to_weight_sharing_format(gv, KFACType.EXPAND, hyper) for gv in g])
        else:
            batch_size = g.shape[0]
            g = grad_to_weight_sharing_format(g, KFACType.EXPAND, hyper).unsqueeze(0)
        a = None
        if "W" in group:
            a = input_to_weight_sharing_format(inputs[0].data.detach(), KFACType.EXPAND, hyper)
            if "b" in group:
                a = torch.cat([a, a.new_ones(*a.shape[:-1], 1)], dim=-1)
        corr = compute_loss_correction(batch_size, self._num_per_example_loss_terms,
                                       self._loss_func.reduction, self._N_data)
        key = tuple(group.values())
        if Qg is None:  # identity bases (GGN diagonal)
            upd = compute_eigenvalue_correction(g, None, a, None).mul_(corr)
        else:
            upd = compute_eigenvalue_correction(g, Qg[key], a, Qa.get(key)).mul_(corr)
        if key in lam:
            lam[key].add_(upd)
        else:
            lam[key] = upd


class HipGGNDiagonalComputer(HipEKFACComputer):
    """Diagonal of the GGN / type-2 Fisher (reference ``computers/ggn_diagonal.py:21-232``) for nets
    whose parameters all belong to ``Linear`` / ``Conv2d`` layers: the EKFAC eigenvalue sweep with
    identity bases, i.e. ``sum_{v,n} (sum_s g_vns a_ns^T)^2`` per layer on the fused
    ``clo_gemm_sqsum_f32`` kernel -- no per-example gradients are materialised."""

    def compute(self) -> dict[str, Tensor]:
        with _use_params(self._model_module, self._params):
            mapping = self.compute_parameter_groups(self._params, self._model_module, False)
            lam = self._eigenvalue_correction(None, None, mapping)
        out: dict[str, Tensor] = {}
        for group in mapping:
            val = lam[tuple(group.values())]
            if "W" in group:
                W = self._params[group["W"]]
                if "b" in group:
                    out[group["b"]] = val[:, -1].contiguous()
                    val = val[:, :-1]
                out[group["W"]] = val.reshape(W.shape)
            else:
                out[group["b"]] = val
        missing = [n for n in self._params if n not in out]
        if missing:
            raise NotImplementedError(f"Parameters outside Linear / Conv2d layers: {missing}.")
        return {n: out[n] for n in self._params}

