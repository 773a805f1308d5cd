"""Build libclo_hip.so (gfx950) in-tree with hipcc.

``python -m curvlinops_amd.csrc.build`` or ``build()``; hipcc cross-compiles without a GPU.
The shared object lands in ``curvlinops_amd/lib/`` so it travels with the source tree.
"""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent
LIB_DIR = CSRC.parent / "lib"
LIB_PATH = LIB_DIR / "libclo_hip.so"
SOURCES = ["gemm.hip", "gemm_v3.hip", "mlp.hip", "mlp_mega.hip", "stream_ops.hip", "linalg.hip", "conv.hip", "gram.hip", "syrk_grouped.hip", "sytrd.hip", "eigh.hip", "eigh_driver.hip", "kron.hip"]
HEADERS = ["clo_common.h", "gemm.h", "persist_gate.h", "mlp_loss.h", "../../include/curvlinops_amd.h"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found; cannot build libclo_hip.so")


def _stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = [CSRC / s for s in SOURCES if (CSRC / s).exists()] + [CSRC / h for h in HEADERS]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source into one shared object for gfx950."""
    if not force and not _stale():
        return LIB_PATH
    LIB_DIR.mkdir(exist_ok=True)
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    obj_dir = LIB_DIR / "obj"
    obj_dir.mkdir(exist_ok=True)

    hdr_time = max((CSRC / h).stat().st_mtime for h in HEADERS)

    def compile_one(src: Path) -> Path:
        obj = obj_dir / (src.stem + ".o")
        # objects are kept between builds (git- and gpurun-ignored): only translation units older than their source
        # or any header are recompiled
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_time):
            return obj
        cmd = [_hipcc(), *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{res.stdout}\n{res.stderr}")
        return obj

    # the translation units are independent: compile them side by side, then link
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=len(srcs)) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
