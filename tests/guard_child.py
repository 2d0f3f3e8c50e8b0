"""A rank of a tiny gloo job for tests/test_guard_cpu.py (not a test module): the stages a guarded
rank of bench.py reports (tools/rank_guard.py), the product's process-group initialisation
(range_amd.dist.init_from_env: timeout, per-attempt store prefix) and one collective per stage, with
faults injected by GUARD_TEST_FAULT:

    hang_overlapped   rank 1 never enters the preflight collective of the OVERLAPPED schedule
                      (attempt 1); the blocking one (RANGE_DIST_BLOCKING=1: attempt 2) is healthy
    hang_always       rank 1 never enters the preflight collective of any attempt
    sleep_past:<s>    rank 1 sleeps <s> seconds before the preflight collective (no guard needed:
                      the process group's timeout must end the job)
    crash_timed       rank 1 raises in the timed stage (behind the preflight: no other attempt)
    fatal             every rank ends with rank_guard.EX_FATAL before the preflight (no GPU, bad arguments)
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    from tools.rank_guard import report_stage
    report_stage("init")
    import torch
    import torch.distributed as dist
    from range_amd.dist import init_from_env
    rank, _, world = init_from_env("gloo")
    fault = os.environ.get("GUARD_TEST_FAULT", "")
    if fault == "fatal":
        from tools.rank_guard import EX_FATAL
        sys.exit(EX_FATAL)
    blocking = os.environ.get("RANGE_DIST_BLOCKING", "0") == "1"
    report_stage("setup")
    dist.barrier()
    report_stage("preflight")
    if rank == 1:
        if fault == "hang_always" or (fault == "hang_overlapped" and not blocking):
            time.sleep(10_000)
        if fault.startswith("sleep_past:"):
            time.sleep(float(fault.split(":")[1]))
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    assert float(t) == world * (world + 1) / 2
    report_stage("timed")
    if fault == "crash_timed" and rank == 1:
        raise RuntimeError("injected failure behind the preflight")
    dist.all_reduce(t)
    report_stage("done")
    if rank == 0:
        print(json.dumps({"schedule": "blocking-fallback" if blocking else "overlapped",
                          "attempt": int(os.environ.get("RANGE_GUARD_ATTEMPT", "1")),
                          "previous_failure": os.environ.get("RANGE_GUARD_PREVIOUS_FAILURE", "")}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
