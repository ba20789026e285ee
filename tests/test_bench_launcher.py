"""bench.py as its own launcher (--gpus N without torchrun), on the CPU box: a rank that never comes back is killed by
the watchdog (--rank-timeout) and every rank's last stderr lines are shown; a rank that fails is reported."""
import os
import subprocess
import sys
import time

from conftest import ROOT


def _run(extra, env_extra, timeout=120):
    env = dict(os.environ, **env_extra)
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(name, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"] + extra, env=env,
                       capture_output=True, text=True, timeout=timeout)
    return p, time.time() - t0


def test_watchdog_kills_a_hung_rank_and_shows_its_stderr():
    p, dt = _run(["--gpus", "2", "--rank-timeout", "4"], {"MK_BENCH_TEST_HANG": "1"})
    assert p.returncode != 0
    assert dt < 60
    assert "did not finish within --rank-timeout 4 s" in p.stderr
    assert "rank 1" in p.stderr and "pretending to hang" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_failed_ranks_are_reported():
    # no GPU here: every rank stops with the reason; the launcher relays it and fails
    p, dt = _run(["--gpus", "2", "--rank-timeout", "100"], {})
    assert p.returncode != 0 and dt < 100
    assert "rank 0" in p.stderr and "rank 1" in p.stderr
    assert "rank(s) failed" in p.stderr or "did not finish" in p.stderr


def test_single_process_refuses_a_launcher():
    env = {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--single-process"], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "--single-process is one process" in p.stderr
