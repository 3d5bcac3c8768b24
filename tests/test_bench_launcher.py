"""CPU: `python bench.py --gpus N` with no launcher around it spawns its own N rank processes (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* as torch.distributed.run sets them), relays rank 0's JSON line and propagates failures.  The
stand-in workload (--selftest-launcher, gloo) exercises the same collectives as the real run: weight broadcast, latent
all_gather, barrier, MAX-over-ranks timing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks_gloo():
    r = run(["--gpus", "2", "--steps", "3", "--selftest-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["selftest"] == "launcher"
    # rank r, item i contributes 16 * (r + 2 i + 1): both ranks' work arrived, in rank order
    assert out["values"] == [16.0 * (r + 2 * i + 1) for r in range(2) for i in range(3)]


def test_external_launcher_env_is_respected_and_mismatch_fails():
    # as the driver launches it: WORLD_SIZE etc. already set -> no spawning; a world/--gpus mismatch is an error
    r = run(["--gpus", "2", "--selftest-launcher"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
    r1 = run(["--gpus", "1", "--steps", "2", "--selftest-launcher"])
    assert r1.returncode == 0 and json.loads(r1.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_failing_rank_propagates():
    # no GPU in the build container: the real (non-selftest) path exits non-zero in every rank, and so does the parent
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "rank(s) failed" in r.stderr


def test_one_rank_dying_early_ends_the_job_promptly():
    """Rank 1 exits before the rendezvous; rank 0 would sit in init_process_group until the backend's timeout.  The
    parent polls every child, terminates the survivor and reports which rank failed — within seconds."""
    import time
    t0 = time.perf_counter()
    r = run(["--gpus", "2", "--steps", "1", "--selftest-launcher"], {"LAVIE_BENCH_SELFTEST_FAIL_RANK": "1"})
    took = time.perf_counter() - t0
    assert r.returncode != 0 and "rank(s) failed" in r.stderr and "(1, 3)" in r.stderr
    assert took < 120, took
