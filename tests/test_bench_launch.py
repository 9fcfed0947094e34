"""bench.py as the driver starts it: `python bench.py --gpus N`, no launcher typed by the caller."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True,
                          text=True, timeout=900)


def test_bench_gpus2_starts_two_ranks_on_one_device():
    """BENCH_DEBUG_ONE_GPU=1: both ranks on device 0, gloo instead of RCCL -- the launcher, the rendezvous, the
    double-buffered steps, the all-gatherv of records and CIGAR words and the max-over-ranks timing, end to end."""
    p = _run("--gpus", "2", "--tasks", "20000", "--steps", "3", "--warmup", "1", BENCH_DEBUG_ONE_GPU="1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["cells_per_step_all_ranks"] >= 2 * 0.95 * d["config"]["cells_per_step_this_rank"]
    assert d["value"] > 0 and d["unit"] == "Gcell/s"


def test_bench_strong_scaling_union_check_two_ranks():
    p = _run("--gpus", "2", "--workload", "hg19mix", "--tasks", "100000", "--strong", "--steps", "2", "--warmup", "1",
             BENCH_DEBUG_ONE_GPU="1")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert "union of 2 shards == single-GPU run" in d["config"]["union_check"]


def test_bench_refuses_more_ranks_than_devices():
    import torch
    n = torch.cuda.device_count() + 1
    p = _run("--gpus", str(n))
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "visible" in p.stderr


def test_bench_forced_dist_one_rank_runs_the_nccl_backend():
    """BENCH_FORCE_DIST=1 python bench.py --gpus 1: the nccl (= RCCL) backend initialised on the device with one rank, every
    step through ResultGatherV on DEVICE tensors -- the count all-gather on RCCL, the stream ordering through wait(), the
    rank's own part equal to what it computed -- and the rank's two ranges sent to itself through the communicator's
    send / receive pair (or the backend's refusal noted): what a box with one GPU can execute of the N > 1 path."""
    p = _run("--gpus", "1", "--tasks", "20000", "--steps", "3", "--warmup", "1", "--no-pcie-pass", BENCH_FORCE_DIST="1")
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["dist_backend"] == "nccl" and d["config"]["forced_dist_one_rank"] is True
    assert "all-gatherv" in d["timed_region"]
    assert d["config"]["loopback"]  # either "... equal" or the backend's refusal, spelled out
    assert d["cpu_baseline"]["value"] > 0
