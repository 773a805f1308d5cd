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
