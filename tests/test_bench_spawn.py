"""CPU (not gpu): `python3 bench.py --gpus N` starts its own N ranks (the driver's scaling run may call it without a
launcher).  On this GPU-less box each rank must get as far as "no GPU visible" -- i.e. the parent spawned one process per
GPU with a rank environment and did not stop at a launcher check -- and the parent must hand back a non-zero exit code.
The N > 1 partitioning itself (the reference's template loop, line2Dup.cpp:1166-1170, cut over ranks) is covered by
tests/test_distributed_gloo.py."""
import os
import re
import subprocess
import sys

from conftest import ROOT


def run_bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_spawns_one_rank_per_gpu_without_a_launcher():
    r = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0  # there is no CPU fallback
    assert "launch with torch.distributed.run" not in r.stderr
    m = re.search(r"spawned 2 ranks \(pids \[(\d+), (\d+)\]\)", r.stderr)
    assert m and m.group(1) != m.group(2), r.stderr
    for rank in (0, 1):
        assert f"bench.py rank {rank} of 2: needs an MI355X: no GPU visible" in r.stderr, r.stderr
    assert re.search(r"rank exit codes: \[1, 1\]", r.stderr), r.stderr
    assert r.stdout == ""  # no JSON line from a run that measured nothing


def test_bench_under_a_launcher_keeps_its_world_size_check():
    r = run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE is 1" in r.stderr and "spawned" not in r.stderr


def test_single_gpu_call_runs_in_process():
    r = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0
    assert "spawned" not in r.stderr
    assert "bench.py rank 0 of 1: needs an MI355X: no GPU visible" in r.stderr


def test_a_rank_that_dies_does_not_leave_the_others_waiting():
    """one rank exits with an error while another would wait (in a rendezvous, a collective) for minutes: the launcher kills
    what it started after a grace period and hands back the failing rank's code"""
    import time

    t0 = time.time()
    r = run_bench(["--gpus", "2"], {"SBM_BENCH_SELFTEST": "rank1_fails_rank0_waits"}, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "was still running 20 s after another rank failed: killed" in r.stderr
    assert 15 < time.time() - t0 < 90
