"""`python bench.py --gpus N` starts its own N ranks: the launcher, the per-rank shares, the collectives (gloo here) and
the JSON line, on a machine without a GPU, with the oracle standing in for the tracer (tests/bench_stub.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT


def _run(extra, env_extra=None, expect_rc=0):
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_TRACER="bench_stub:StubTracer",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, os.environ.get("PYTHONPATH", "")]),
               RT_BENCH_SPAWN_TIMEOUT="240")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cylinder_100x60", "--steps", "2", "--warmup", "1",
                         "--min-seconds", "0"] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert pr.returncode == expect_rc, pr.stderr[-2000:]
    return pr


@pytest.mark.parametrize("split", ["tile", "sample"])
def test_bench_starts_its_own_ranks(split):
    """--gpus 2 with no launcher around it: two ranks run, rank 0 reports n_gpus = 2 and the rays of both."""
    one = json.loads(_run(["--gpus", "1"]).stdout.strip().splitlines()[-1])
    two = json.loads(_run(["--gpus", "2", "--split", split]).stdout.strip().splitlines()[-1])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["split"] == split
    assert two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "strong"
    # the work is the same render either way: a tile split traces exactly the single rank's rays, a sample split the same
    # number of camera rays with other random numbers
    rays1, rays2 = one["value"] * one["ms_per_step"], two["value"] * two["ms_per_step"]
    if split == "tile":
        assert abs(rays1 - rays2) < 1e-6 * rays1
    else:
        assert abs(rays1 - rays2) < 0.05 * rays1
    assert two["config"]["rays_per_step_per_gpu"] < one["config"]["rays_per_step_per_gpu"]


def test_bench_refuses_a_world_that_does_not_match():
    """Launched as one rank of three while asked for --gpus 2: an error, not a mislabelled line."""
    pr = _run(["--gpus", "2"], env_extra={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, expect_rc=1)
    assert "WORLD_SIZE=3" in pr.stderr and not pr.stdout.strip()


def test_bench_refuses_more_gpus_than_are_visible():
    """With the real tracer and RCCL, --gpus 2 on a machine that shows fewer than two devices exits non-zero before any
    rank starts (this container shows none)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two or more GPUs are visible here")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RT_BENCH_BACKEND", "RT_BENCH_TRACER"):
        env.pop(k, None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert pr.returncode == 2 and "HIP device(s) visible" in pr.stderr and not pr.stdout.strip()
