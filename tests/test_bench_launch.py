"""`bench.py --gpus N` must start its N ranks itself (the driver's plain command) and refuse
silently wrong world sizes; the N = 2 dry run shares one device over gloo."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_world_size_mismatch_is_an_error():
    # a launcher that started 1 rank for --gpus 2 must not produce an n_gpus = 1 line
    res = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"], {"WORLD_SIZE": "1", "RANK": "0"}, 300)
    assert res.returncode != 0
    assert "WORLD_SIZE=1" in res.stderr
    assert not res.stdout.strip()


@pytest.mark.gpu
def test_self_launch_two_ranks_on_one_device():
    res = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras"], {"CLO_BENCH_BACKEND": "gloo"}, 900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3
    assert rec["value"] > 0 and rec["config"]["parallelism"].startswith("dp2")


@pytest.mark.gpu
def test_self_launch_refuses_more_ranks_than_devices():
    import torch

    n = torch.cuda.device_count() + 1
    res = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-extras"], {}, 300)
    assert res.returncode != 0 and "device(s) visible" in res.stderr
