"""First contact of a multi-rank job must be BOUNDED (round-5 verdict, item 1): a rank that never
arrives costs at most the process group's timeout, never the job's whole budget; a hang of the
overlapped schedule is answered by fresh rank processes on the blocking schedule
(tools/rank_guard.py); whatever happens, every rank process ends, non-zero when there is no result.

Two-rank gloo jobs under torch.distributed.run, exactly as the driver launches bench.py; the ranks
are tests/guard_child.py (the product's init_from_env + the stage reports of bench.py's ranks) with
injected faults."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(REPO, "tests", "guard_child.py")
GUARD = os.path.join(REPO, "tools", "rank_guard.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, prog_args, env_extra, timeout=240):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + prog_args
    env = dict(os.environ)
    env.update(env_extra)
    env.pop("RANGE_GUARD_FD", None)
    t0 = time.monotonic()
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr, time.monotonic() - t0


def _result_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_healthy_job_runs_once_on_the_overlapped_schedule():
    rc, out, err, dt = _launch(2, [GUARD, "--preflight-timeout", "20", "--", sys.executable, CHILD], {})
    assert rc == 0, err[-2000:]
    assert _result_line(out) == {"schedule": "overlapped", "attempt": 1, "previous_failure": ""}


def test_hang_of_the_overlapped_schedule_falls_back_to_fresh_blocking_ranks():
    """Rank 1 never enters the preflight collective of attempt 1: its guard ends it at the preflight
    deadline, rank 0's child is thrown out of (or ended in) the same collective, BOTH guards start
    fresh children with RANGE_DIST_BLOCKING=1 under a new store prefix - and the job has a result."""
    rc, out, err, dt = _launch(2, [GUARD, "--preflight-timeout", "6", "--", sys.executable, CHILD],
                               {"GUARD_TEST_FAULT": "hang_overlapped", "RANGE_DIST_TIMEOUT_S": "30"})
    assert rc == 0, err[-3000:]
    res = _result_line(out)
    assert res["schedule"] == "blocking-fallback" and res["attempt"] == 2
    assert "attempt 1 (overlapped)" in res["previous_failure"]
    assert dt < 120, dt


def test_a_rank_that_never_arrives_ends_every_rank_nonzero_and_soon():
    """No schedule helps: both attempts hang.  Every rank process ends, the launcher reports failure,
    well inside the 180 s the verdict allows (and the 600 s of a benchmark run)."""
    rc, out, err, dt = _launch(2, [GUARD, "--preflight-timeout", "5", "--", sys.executable, CHILD],
                               {"GUARD_TEST_FAULT": "hang_always", "RANGE_DIST_TIMEOUT_S": "30"})
    assert rc != 0
    assert not [ln for ln in out.splitlines() if ln.startswith("{")]
    assert dt < 180, dt


def test_process_group_timeout_alone_bounds_a_late_rank():
    """Without any guard: rank 1 sleeps past the process group's timeout (init_from_env: 8 s here,
    120 s by default instead of torch's 600) - rank 0's collective gives up, the launcher ends rank 1,
    every rank exits non-zero in well under 180 s."""
    rc, out, err, dt = _launch(2, [CHILD], {"GUARD_TEST_FAULT": "sleep_past:100", "RANGE_DIST_TIMEOUT_S": "8"})
    assert rc != 0
    assert dt < 90, dt


def test_failure_behind_the_preflight_is_not_retried():
    rc, out, err, dt = _launch(2, [GUARD, "--preflight-timeout", "20", "--", sys.executable, CHILD],
                               {"GUARD_TEST_FAULT": "crash_timed", "RANGE_DIST_TIMEOUT_S": "20"})
    assert rc != 0
    assert "the preflight had passed" in err
    assert "starting a fresh child" not in err
    assert dt < 120, dt


def test_a_configuration_error_is_not_answered_with_another_schedule():
    rc, out, err, dt = _launch(2, [GUARD, "--", sys.executable, CHILD], {"GUARD_TEST_FAULT": "fatal"})
    assert rc != 0 and "no schedule cures it" in err and "starting a fresh child" not in err
    assert dt < 60, dt


def test_default_timeout_is_two_minutes_and_env_overridable(monkeypatch):
    from range_amd import dist as rdist
    monkeypatch.delenv("RANGE_DIST_TIMEOUT_S", raising=False)
    assert rdist.dist_timeout_s() == 120.0
    monkeypatch.setenv("RANGE_DIST_TIMEOUT_S", "45")
    assert rdist.dist_timeout_s() == 45.0
    assert rdist.dist_timeout_s(7) == 7.0
