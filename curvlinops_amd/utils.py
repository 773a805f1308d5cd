"""Small host-side helpers shared by the operators (device/dtype inference, comparisons,
functional calls).  Behaviour mirrors the helpers of the reference's ``curvlinops/utils.py``
(e.g. ``allclose_report`` :173-215, ``_infer_device`` :21-36, ``split_list`` :147-170) so that
error types and messages line up; the code is written for this package."""

from __future__ import annotations

from collections.abc import Callable, Iterable, Iterator
from contextlib import contextmanager

import numpy
import torch
from torch import Tensor
from torch.func import functional_call
from torch.nn import CrossEntropyLoss, Module


def infer_device(objects: Iterable) -> torch.device:
    """Common device of tensors/operators; ``RuntimeError`` if they disagree."""
    found = {o.device for o in objects}
    if len(found) != 1:
        raise RuntimeError(f"Expected single device, got {found}.")
    return found.pop()


def infer_dtype(objects: Iterable) -> torch.dtype:
    """Common dtype of tensors/operators; ``RuntimeError`` if they disagree."""
    found = {o.dtype for o in objects}
    if len(found) != 1:
        raise RuntimeError(f"Expected single dtype, got {found}.")
    return found.pop()


def allclose_report(a: Tensor | numpy.ndarray, b: Tensor | numpy.ndarray, rtol: float = 1e-5,
                    atol: float = 1e-8) -> bool:
    """``allclose`` that prints a short diagnosis of the mismatch when it fails."""
    a = torch.as_tensor(a)
    b = torch.as_tensor(b, device=a.device)
    ok = bool(torch.allclose(a, b, rtol=rtol, atol=atol))
    if not ok:
        bad = ~torch.isclose(a, b, rtol=rtol, atol=atol)
        n_bad = int(bad.sum())
        idx = bad.nonzero()[:10]
        for i in idx:
            t = tuple(i.tolist())
            print(f"at index {list(t)}: {a[t].item():.5e} != {b[t].item():.5e}")
        print(f"Abs max: {a.abs().max().item():.5e} vs. {b.abs().max().item():.5e}.")
        print(f"Non-close entries: {n_bad} / {a.numel()}. rtol = {rtol}, atol = {atol}.")
    return ok


def split_list(items: list | tuple, sizes: list[int]) -> list[list]:
    """Cut ``items`` into consecutive sub-lists of the given sizes."""
    if len(items) != sum(sizes):
        raise ValueError(
            f"List to be split has length {len(items)}, but requested sub-list with a total"
            f" of {sum(sizes)} entries."
        )
    out, pos = [], 0
    for n in sizes:
        out.append(list(items[pos : pos + n]))
        pos += n
    return out


def assert_is_square(A) -> int:
    if len(A.shape) != 2 or A.shape[0] != A.shape[1]:
        raise ValueError(f"Operator must be square. Got shape {A.shape}.")
    return int(A.shape[0])


def assert_matvecs_subseed_dim(A, num_matvecs: int) -> None:
    if any(num_matvecs >= d for d in A.shape):
        raise ValueError(f"num_matvecs ({num_matvecs}) must be less than A's size ({A.shape}).")


def assert_divisible_by(num: int, divisor: int, name: str) -> None:
    if num % divisor != 0:
        raise ValueError(f"{name} ({num}) must be divisible by {divisor}.")


def make_functional_call(module: Module) -> Callable[..., Tensor]:
    """``(params_dict, *inputs) -> module(*inputs)`` with ``params_dict`` overriding the
    module's own parameters (buffers and frozen parameters fall through)."""

    def call(params: dict[str, Tensor], *inputs) -> Tensor:
        return functional_call(module, params, inputs)

    return call


def make_functional_loss(loss_func: Module) -> Callable[[Tensor, tuple], Tensor]:
    """``(prediction, loss_args) -> loss`` for a criterion module."""

    def c(prediction: Tensor, loss_args: tuple) -> Tensor:
        return functional_call(loss_func, {}, (prediction, *loss_args))

    return c


def flatten_output_and_labels(output: Tensor, y: Tensor, loss_func: Module) -> tuple[Tensor, Tensor]:
    """Fold weight-sharing axes into the batch axis: CE ``(b, c, ...) -> ((b ...), c)`` and
    labels ``(b, ...) -> (b ...)``; other losses ``(b, ..., c) -> ((b ...), c)`` (reference
    ``utils.py:352-362``, ``computers/_base.py:258-266``)."""
    if isinstance(loss_func, CrossEntropyLoss):
        return output.movedim(1, -1).flatten(0, -2), y.flatten()
    return output.flatten(0, -2), y.flatten(0, -2)


@contextmanager
def enable_requires_grad(tensors: list[Tensor]) -> Iterator[None]:
    before = [t.requires_grad for t in tensors]
    for t in tensors:
        t.requires_grad_(True)
    try:
        yield
    finally:
        for t, rg in zip(tensors, before):
            t.requires_grad_(rg)


def seed_generator(generator: torch.Generator | None, dev: torch.device, seed: int) -> torch.Generator:
    if generator is None or generator.device != dev:
        generator = torch.Generator(device=dev)
    generator.manual_seed(seed)
    return generator


def is_native_tensor(t: Tensor) -> bool:
    """True if ``t`` can be handed to the HIP kernels (fp32 on a GPU)."""
    return t.is_cuda and t.dtype == torch.float32


_SIDE_STREAMS: dict = {}


def side_stream(device: torch.device, index: int) -> "torch.cuda.Stream":
    """The package's ONE pool of persistent worker streams per device (eigensolver / inverse workers, the Kronecker
    block pool, the KFAC factor stream all draw from it by index).  Persistent: the caching allocator keeps freed blocks
    PER STREAM, so workers that made a fresh stream per call could never reuse the (GB-sized) workspaces of the previous
    call (inverse time 15 <-> 58 ms from run to run).  Shared: HIP maps streams onto a handful of hardware queues in
    creation order, and streams that share a queue serialise -- a dozen private pools made WHICH ones collide depend on
    the order in which the subsystems were first used."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]
