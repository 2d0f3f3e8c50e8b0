"""A guard around ONE rank of a multi-process job: the rank's work runs in a fresh CHILD process, the
guard (this process: no torch, no GPU) watches the stages the child reports and, when the child's
first contact with its peers fails or hangs, starts a fresh child with a more conservative schedule.

Why: the overlapped, chunked schedule of range_amd/dist.py has run over gloo, over torch's in-process
backend and over RCCL with one rank - with more than one RCCL rank it has never executed (no
multi-GPU hardware in six rounds).  Its first run will be the driver's scaling benchmark, whose
budget a hang would eat whole and leave no record.  A process whose collective hangs cannot be
repaired from inside (its GPU stream is stuck behind the collective's kernel), and a process that has
touched the GPU must never be re-exec'ed: the only clean recovery is to end it and start another.

    guard (rank process started by torchrun)          child (fresh python, the rank's work)
      attempt 1: env as given                           stage init -> setup -> preflight -> timed -> done
      attempt 2: + RANGE_DIST_BLOCKING=1                (written to the pipe RANGE_GUARD_FD names)
                 + RANGE_DIST_ATTEMPT=2 (a store prefix of its own: dist.init_from_env)

Rules (every guard of the job applies them on its own; they converge without talking to each other):
  * a child that ends with 0 ends the guard with 0;
  * a child that fails - any exit code, or killed here because a stage outlived its deadline - BEFORE
    it reported the stage ``timed`` sends the guard to the next attempt (its peers are then stuck in,
    or thrown out of, the same collective: their guards see the same within the process group's
    timeout or their own stage deadline, whichever comes first);
  * a failure at or behind ``timed`` (the preflight had passed), a child that ends with EX_FATAL (78: no
    GPU, bad arguments - nothing a schedule cures), a failed last attempt, or no time left for another
    attempt ends the guard non-zero: torchrun then ends the other ranks;
  * SIGTERM / SIGINT (torchrun tearing the job down) kill the child and end the guard; a guard that
    dies without warning takes its child with it (PR_SET_PDEATHSIG).

CLI (tests/test_guard_cpu.py):  python tools/rank_guard.py [--preflight-timeout S] ... -- prog args...
"""
from __future__ import annotations

import argparse
import ctypes
import os
import select
import signal
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple

#: exit code of a child that itself concluded "this schedule does not work here, try the next one"
EX_RETRY = 75
#: exit code of a child whose failure no other schedule can cure (no GPU, bad arguments): never retried
EX_FATAL = 78

#: the stages a child reports, in order (``report_stage``); the guard's deadlines hang on them
STAGES = ("init", "setup", "preflight", "timed", "done")

DEFAULT_DEADLINES = {
    # seconds a child may stay IN a stage.  "start" = until its first report (python start-up and
    # the first ``import torch`` of a fresh box: 1-2 minutes - the child reports "init" only behind its
    # imports); "init" = process-group rendezvous + communicator; "setup" = bank / engine construction; "preflight" = the first steps over the
    # collectives, blocking then overlapped - the stage a hang is expected in, if anywhere
    "start": 240.0, "init": 180.0, "setup": 180.0, "preflight": 60.0, "timed": 300.0, "done": 60.0,
}


def report_stage(name: str) -> None:
    """Called by the CHILD: tell the guard (if there is one) which stage begins."""
    fd = os.environ.get("RANGE_GUARD_FD")
    if not fd:
        return
    try:
        os.write(int(fd), f"stage {name}\n".encode())
    except OSError:
        pass


def guarded() -> bool:
    return bool(os.environ.get("RANGE_GUARD_FD"))


def _die_with_parent():
    # (runs in the child between fork and exec: the guard has not touched a GPU, neither has this)
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)      # PR_SET_PDEATHSIG
    except Exception:  # noqa: BLE001
        pass


def _log(msg: str) -> None:
    rank = os.environ.get("RANK", "?")
    print(f"[rank_guard rank {rank}] {msg}", file=sys.stderr, flush=True)


def guard_rank(argv: Sequence[str], attempts: Sequence[Tuple[str, Dict[str, str]]],
               deadlines: Optional[Dict[str, float]] = None, total_timeout: float = 540.0) -> int:
    """Run ``argv`` as a child once per attempt ``(name, extra env)`` until one succeeds (rules in the
    module docstring).  Returns the exit code for this rank process."""
    dl = dict(DEFAULT_DEADLINES)
    dl.update(deadlines or {})
    t_job = time.monotonic()
    child: List[Optional[subprocess.Popen]] = [None]

    def on_signal(signum, _frame):
        p = child[0]
        if p is not None and p.poll() is None:
            p.kill()
        _log(f"signal {signum}: child ended, leaving")
        os._exit(128 + signum)

    old = {s: signal.signal(s, on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    reason = ""
    try:
        for i, (name, extra) in enumerate(attempts):
            r, w = os.pipe()
            env = dict(os.environ)
            env.update(extra)
            env["RANGE_GUARD_FD"] = str(w)
            env["RANGE_GUARD_ATTEMPT"] = str(i + 1)
            env["RANGE_GUARD_ATTEMPT_NAME"] = name
            if i:
                env["RANGE_DIST_ATTEMPT"] = str(i + 1)
                env["RANGE_GUARD_PREVIOUS_FAILURE"] = reason
            proc = child[0] = subprocess.Popen(list(argv), env=env, pass_fds=(w,), preexec_fn=_die_with_parent)
            os.close(w)
            stage, t_stage = "start", time.monotonic()
            buf = b""
            killed = None
            pipe_open = True

            def drain(block_s: float):
                """Read what the child has reported; returns False once the pipe is closed."""
                nonlocal buf, stage, t_stage
                ready, _, _ = select.select([r], [], [], block_s)
                if not ready:
                    return True
                data = os.read(r, 4096)
                if not data:
                    return False
                buf += data
                while b"\n" in buf:
                    line, buf = buf.split(b"\n", 1)
                    parts = line.decode(errors="replace").split()
                    if len(parts) == 2 and parts[0] == "stage" and parts[1] in STAGES:
                        stage, t_stage = parts[1], time.monotonic()
                return True

            while True:
                if pipe_open:
                    pipe_open = drain(0.25)
                else:
                    time.sleep(0.25)
                if proc.poll() is not None:
                    while pipe_open:               # (the last reports of a child that has ended)
                        pipe_open = drain(0.0) and bool(select.select([r], [], [], 0.0)[0])
                    break
                now = time.monotonic()
                if now - t_stage > dl[stage]:
                    killed = f"stage '{stage}' outlived its {dl[stage]:.0f} s deadline"
                elif now - t_job > total_timeout:
                    killed = f"the job's {total_timeout:.0f} s are over (stage '{stage}')"
                if killed:
                    proc.kill()
                    proc.wait()
                    break
            os.close(r)
            rc = proc.wait()
            child[0] = None
            if rc == 0 and not killed:
                return 0
            reason = (killed or f"exit code {rc}") + f" in attempt {i + 1} ({name})"
            passed = STAGES.index(stage) >= STAGES.index("timed") if stage in STAGES else False
            left = total_timeout - (time.monotonic() - t_job)
            if passed or rc == EX_FATAL or i + 1 == len(attempts) or left < 60.0:
                _log(f"{reason}: giving up" + (" (the preflight had passed: no other schedule would have been timed)" if passed else
                                               " (a configuration error: no schedule cures it)" if rc == EX_FATAL else ""))
                return rc if rc > 0 else 1
            _log(f"{reason}: starting a fresh child for attempt {i + 2} ({attempts[i + 1][0]})")
        return 1
    finally:
        for s, h in old.items():
            signal.signal(s, h)


#: the attempts of a rank of the row-sharded job: the overlapped schedule, then every collective blocking
SHARDED_ATTEMPTS = (("overlapped", {}), ("blocking-fallback", {"RANGE_DIST_BLOCKING": "1"}))


def main() -> int:
    ap = argparse.ArgumentParser()
    for s in DEFAULT_DEADLINES:
        ap.add_argument(f"--{s}-timeout", type=float, default=None)
    ap.add_argument("--total-timeout", type=float, default=540.0)
    ap.add_argument("--single-attempt", action="store_true")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        ap.error("no command")
    dls = {s: getattr(a, f"{s}_timeout") for s in DEFAULT_DEADLINES if getattr(a, f"{s}_timeout") is not None}
    attempts = SHARDED_ATTEMPTS[:1] if a.single_attempt else SHARDED_ATTEMPTS
    return guard_rank(cmd, attempts, dls, a.total_timeout)


if __name__ == "__main__":
    sys.exit(main())
