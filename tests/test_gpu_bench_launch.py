"""`python bench.py --gpus N` starts its N ranks itself (as the reference's RenderEngine spawns its
own workers, integrator.cpp:361-378, and sums their canvases, :469-487).  Run here the way the
driver runs it -- directly, no launcher -- as a two-rank rehearsal on the one GPU of the test box."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--level", "5", "--width", "320",
           "--height", "200", "--spp-per-step", "8", "--strong-spp", "16", "--no-cpu-baseline"] + list(extra)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    return res


def test_bench_gpus_2_launches_two_ranks():
    res = _bench("--gpus", "2", "--rehearse")
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert out["film_check"] is True                         # every pixel got both ranks' passes
    assert out["strong_scaling"]["spp_per_rank"] == [8, 8] and out["strong_scaling"]["film_check"] is True
    assert out["strong_scaling"]["n_gpus"] == 2
    assert out["camera_samples"] == 2 * 8 * 320 * 200       # weak scaling: both ranks' samples


def test_bench_rank_failure_is_reported():
    """A rank that dies takes the whole launch down with a non-zero status and no JSON line."""
    res = _bench("--gpus", "2", "--rehearse", env={"PT_HIP_LIB": "/nonexistent/libpt_hip.so"})
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]


def test_bench_collectives_over_rccl_with_one_rank():
    """`--dist-one-rank`: the process group is set up with the `nccl` backend (RCCL) for the single rank
    and the film reduce, the sums and maxima over ranks and the barriers all go through it -- the calls
    the N-GPU run makes, on the one GPU of the test box."""
    res = _bench("--gpus", "1", "--dist-one-rank")
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["film_check"] is True and out["strong_scaling"]["film_check"] is True
    assert "RCCL" in out["config"]["parallelism"]
    assert out["camera_samples"] == 8 * 320 * 200
