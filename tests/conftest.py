"""Shared pytest configuration: the ``gpu`` marker and golden-vector loading."""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without a GPU every ``gpu`` test is skipped (a plain ``pytest tests`` on a CPU-only host must not fail)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP GPU available")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name: str) -> dict[str, dict[str, np.ndarray]]:
    """Load ``tests/golden/<name>.npz`` into {case: {key: array}}."""
    flat = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    cases: dict[str, dict[str, np.ndarray]] = {}
    for k in flat.files:
        case, key = k.split("/", 1)
        cases.setdefault(case, {})[key] = flat[k]
    return cases


def mlp_case_tensors(rec):
    """Unpack a golden MLP record -> (dims, acts, bias flags, loss, reduction, Ws, bs, data)."""
    dims = [int(d) for d in rec["dims"]]
    acts = [str(a) for a in rec["acts"]]
    bias = [bool(b) for b in rec["bias"]]
    loss, red = str(rec["loss"]), str(rec["reduction"])
    names = sorted((k for k in rec if k.startswith("param:")), key=lambda k: int(k.split(":")[1].split(".")[0]))
    Ws, bs = [], []
    lin_ids = sorted({int(k.split(":")[1].split(".")[0]) for k in names})
    for i in lin_ids:
        Ws.append(rec[f"param:{i}.weight"])
        bs.append(rec.get(f"param:{i}.bias"))
    data = [(rec[f"X{i}"], rec[f"y{i}"]) for i in range(int(rec["num_batches"]))]
    return dims, acts, bias, loss, red, Ws, bs, data


@pytest.fixture(scope="session")
def golden_mlp():
    return load_golden("mlp_curvature")


@pytest.fixture(scope="session")
def golden_linops():
    return load_golden("linops")
