"""`python bench.py --gpus N` WITHOUT a launcher around it -- the form the driver uses -- must start its own ranks
(bench.spawn_ranks: torch.distributed.run as a child process on a free port, rank 0's JSON line relayed, the child's exit code
returned) instead of dying on an assertion.  --dry-launch runs that plumbing without a GPU: every rank joins a gloo group and
rank 0 reports how many ranks the collectives saw.  The real N-rank flow on one GPU is tests/test_gpu_distributed.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bare(args, timeout=500, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n", [2, 8])
def test_bare_multi_gpu_command_starts_its_own_ranks(n):
    """BASELINE config 4's command line, bare: `bench.py --gpus 8 --total-images 50000 --steps 20 --warmup 5`."""
    r = _bare(["--gpus", str(n), "--total-images", "50000", "--steps", "20", "--warmup", "5", "--dry-launch"],
              env={"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly one JSON line on stdout, rank 0's
    d = json.loads(lines[0])
    assert d["dry_launch"] is True and d["n_gpus"] == n and d["ranks_seen"] == n and d["scaling"] == "strong"
    assert sorted(x[0] for x in d["devices"]) == list(range(n))
    # strong scaling keeps the forward's shape: 256 images per batch at EVERY N (so that t_1 and t_N time the same kernels), the
    # job is ceil(50000 / 256) = 196 batches dealt round-robin, and --steps is derived: the batches of the busiest rank
    assert d["config"]["batch"] == 256 and d["config"]["batches_total"] == 196 and d["config"]["images_total"] == 196 * 256
    assert d["steps"] == -(-196 // n) and d["config"]["images_per_gpu"] == 256 * -(-196 // n)
    assert d["config"]["parallelism"] == "dp%d" % n
    assert "torch.distributed.run" in r.stderr             # the parent said what it started


@pytest.mark.timeout(300)
def test_strong_scaling_at_one_gpu_runs_the_same_batch_as_at_eight():
    """The N = 1 leg of BASELINE config 4's curve: same 196 batches of 256 images, all on the one rank (no launcher, no group)."""
    r = _bare(["--gpus", "1", "--total-images", "50000", "--steps", "20", "--warmup", "5", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["config"]["batch"] == 256 and d["steps"] == 196
    assert d["config"]["batches_total"] == 196 and d["config"]["images_per_gpu"] == 196 * 256


@pytest.mark.timeout(300)
def test_weak_scaling_default_is_the_drivers_configuration():
    r = _bare(["--gpus", "1", "--steps", "20", "--warmup", "5", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][-1])
    assert d["scaling"] == "weak" and d["config"]["batch"] == 256 and d["steps"] == 20 and d["config"]["images_total"] == 5120


@pytest.mark.timeout(300)
def test_bare_multi_gpu_command_on_a_box_with_too_few_gpus_says_so_and_returns():
    """Without --dry-launch and without the gloo override the bare form needs N devices: it must say so and return a
    non-zero code at once (no rank started, no GPU call, no hang)."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has 8 GPUs")
    r = _bare(["--gpus", "8", "--steps", "2", "--warmup", "1"], timeout=200, env={"FQ_BENCH_BACKEND": "nccl"})
    assert r.returncode == 2 and "GPU(s)" in r.stderr and not r.stdout.strip()


@pytest.mark.timeout(300)
def test_rank_count_mismatch_is_reported_not_asserted():
    r = _bare(["--gpus", "4", "--dry-launch"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29688"})
    # a launcher that started one rank for --gpus 4: the dry launch still reports what the collectives saw
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 1
