"""bench.py --gpus N outside torch.distributed.run: the process becomes a launcher that starts N rank processes itself
(VERDICT r3 item 1).  CPU tests of that path: rank environment, one clean JSON line on stdout, failure propagation."""
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "launcher_child.py")


def _run(mode, n=2, timeout=None):
    sys.path.insert(0, ROOT)
    import bench
    out, err = io.StringIO(), io.StringIO()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    rc = bench.spawn_ranks(n, [sys.executable, CHILD, mode], env=env, out=out, err=err, timeout=timeout)
    return rc, out.getvalue(), err.getvalue()


def test_launcher_gives_up_when_every_rank_blocks(monkeypatch):
    """all ranks blocked (no rank ever exits non-zero) used to hang the launcher for good.  With a deadline it
    terminates the ranks, escalates to kill() for the ones that ignore SIGTERM, dumps their output and fails with 124."""
    import time
    t0 = time.monotonic()
    rc, out, err = _run("hang", 2, timeout=8.0)
    assert rc == 124 and out.strip() == ""
    assert "no result after 8 s" in err and "[rank 0] [noise] rank 0 is stuck" in err and "[rank 1] [noise] rank 1 is stuck" in err
    assert time.monotonic() - t0 < 60


def test_launcher_starts_n_ranks_and_relays_one_clean_line():
    rc, out, err = _run("ok", 2)
    assert rc == 0, err
    lines = out.strip().splitlines()
    assert len(lines) == 1                                   # stdout carries the result line and nothing else
    d = json.loads(lines[0])
    assert d["metric"] == "launcher_selftest" and d["n_gpus"] == 2 and d["dist_world_size"] == 2
    assert d["value"] == 3.0                                 # 1 + 2: both ranks took part in the all-reduce
    assert "[rank 0] [noise]" in err and "[rank 1] [noise]" in err and "not_the_line" in err


def test_launcher_fails_when_a_rank_fails():
    rc, out, err = _run("fail", 2)
    assert rc == 7 and out.strip() == ""
    assert "rank 1 exited with code 7" in err


def test_launcher_fails_without_a_result_line():
    rc, out, err = _run("silent", 2)
    assert rc == 1 and out.strip() == "" and "no result line" in err


def test_bench_cli_takes_the_launcher_path_and_propagates_failure():
    """`python bench.py --gpus 2` with no WORLD_SIZE: on a box without a GPU the rank processes cannot run the HIP path — they
    fail loudly, and so does the launcher (non-zero exit, nothing on stdout).  The same command on the GPU box prints one
    line with n_gpus = 2 (tests/test_gpu_sharded_step.py)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check of the failure path")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-ahds",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "bench.py launcher: rank" in r.stderr
