"""bench.py's multi-rank launch is bounded and self-diagnosing (round-5 verdict, item 1): two gloo rank
processes sharing the box's GPU run the N > 1 code path of the bench - rank guards, preflight
(blocking == overlapped, bit for bit), the fall-back to the blocking schedule on a mismatch (same
process) and on a hang (fresh rank processes) - and every outcome is a line that says what happened.
(RCCL needs one GPU per rank; what the guard and the preflight do does not depend on the backend.)"""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, timeout=500):
    env = dict(os.environ)
    env.update({"RANGE_DIST_BACKEND": "gloo"})
    env.update(extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RANGE_GUARD_FD"):
        env.pop(k, None)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--queries", "2048", "--bank", "range_db_med", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=timeout)
    dt = time.monotonic() - t0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    return p, (json.loads(lines[0]) if lines else None), dt


def test_two_rank_bench_preflight_passes_on_the_overlapped_schedule():
    p, line, dt = _bench({})
    assert p.returncode == 0 and line is not None, p.stdout[-2000:] + p.stderr[-3000:]
    d = line["dist"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["schedule"] == "overlapped"
    pf = d["preflight"]
    assert pf["bit_identical"] is True and pf["ranks_agree"] is True and pf["chunks_overlapped"] == 2
    assert pf["blocking_ms"] > 0 and pf["overlapped_ms"] > 0
    assert d["guard"] == {"guarded": True, "attempt": 1, "previous_failure": None}
    assert d["timeout_s"] == 120.0
    assert line["parity_max_abs"] < 1e-4


def test_two_rank_bench_mismatch_falls_back_to_blocking_in_the_same_process():
    p, line, dt = _bench({"RANGE_BENCH_INJECT": "mismatch_overlapped"})
    assert p.returncode == 0 and line is not None, p.stdout[-2000:] + p.stderr[-3000:]
    d = line["dist"]
    assert d["schedule"] == "blocking-fallback" and d["preflight"]["ranks_agree"] is False
    assert d["guard"]["attempt"] == 1
    assert line["parity_max_abs"] < 1e-4          # (the timed result of the blocking schedule is checked as any other)


def test_two_rank_bench_hang_is_answered_by_fresh_blocking_ranks():
    p, line, dt = _bench({"RANGE_BENCH_INJECT": "hang_overlapped", "RANGE_BENCH_PREFLIGHT_TIMEOUT_S": "25"})
    assert p.returncode == 0 and line is not None, p.stdout[-2000:] + p.stderr[-3000:]
    d = line["dist"]
    assert d["schedule"] == "blocking-fallback" and d["guard"]["attempt"] == 2
    assert "attempt 1 (overlapped)" in d["guard"]["previous_failure"]
    assert d["preflight"]["overlapped_ms"] is None
    assert line["parity_max_abs"] < 1e-4
    assert dt < 400, dt
