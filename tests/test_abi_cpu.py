"""The C-ABI library loads without a GPU and exports every symbol include/curvlinops_amd.h
declares (no compute calls here)."""

import re
from pathlib import Path

from curvlinops_amd import _hip

HEADER = Path(__file__).resolve().parent.parent / "include" / "curvlinops_amd.h"


def test_library_exports_every_declared_symbol():
    lib = _hip.load()
    text = HEADER.read_text()
    declared = set(re.findall(r"\b(clo_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_hip.exported_symbols()), declared ^ set(_hip.exported_symbols())
    assert lib.clo_version() >= 100


def test_argument_validation_without_gpu():
    lib = _hip.load()
    # negative sizes are rejected before any HIP call
    assert lib.clo_axpby_f32(None, None, -1, 1.0, 0.0, None) == -1
    assert b"negative" in lib.clo_last_error()
    assert lib.clo_gemm_suggest_splitk(128, 128, 64, 1) == 1
    assert lib.clo_gemm_suggest_splitk(27, 27, 500000, 1) > 1
    assert lib.clo_mlp_bwd_ws_floats(8, 2688, 2688) > 0


def test_kmajor_layout_predicate():
    """`canonical.is_kmajor`: the stride signature by which the canonical converters and the Kronecker blocks recognise a
    `[n, K]` view of a contiguous `[K, n]` array (host logic of the fused pack / unpack, no GPU needed)."""
    import torch

    from curvlinops_amd.canonical import is_kmajor

    assert is_kmajor(torch.empty(5, 12).T)                    # [12, 5] view of [5, 12]
    assert not is_kmajor(torch.empty(12, 5))                  # plain K-trailing
    assert not is_kmajor(torch.empty(12, 1))                  # single vectors are never K-major
    assert not is_kmajor(torch.empty(1, 12).T[:, :1])
    assert not is_kmajor(torch.empty(5, 24).T[::2])           # strided rows
    assert not is_kmajor(torch.empty(3, 4, 5))
