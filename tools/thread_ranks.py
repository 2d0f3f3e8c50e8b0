"""W ranks of a torch.distributed job as W THREADS of this process (test / rehearsal infrastructure:
tests/test_gpu_world8.py, tests/test_dist_cpu.py, ``RANGE_DIST_BACKEND=threads python bench.py --gpus 8``).

A GPU box of this pool admits at most 6 processes on its card, and the north-star world size is 8:
the ranks of the world-8 GPU tests are therefore threads, each with its own engine context on
cuda:0, talking through torch's in-process "threaded" process group
(torch.testing._internal.distributed.multi_threaded_pg: every collective is carried out by the
last thread to arrive, as plain tensor copies on the device - no host staging, no second process).
What it exercises is everything of range_amd/dist.py and range_amd/save.py above the backend:
shapes, chunking, buffer re-use, sub-groups, byte counters, the order of the collectives.  What it
cannot reach is RCCL itself (tests/test_gpu_rccl.py)."""
import threading
import traceback

import torch
import torch.distributed as dist


def threaded_backend_available() -> bool:
    try:
        from torch.testing._internal.distributed import multi_threaded_pg  # noqa: F401
        return hasattr(torch._C._distributed_c10d, "_set_thread_isolation_mode")
    except Exception:  # noqa: BLE001
        return False


def run_rank_threads(world: int, fn, *args, timeout: float = 900.0) -> dict:
    """Run ``fn(rank, world, *args)`` on ``world`` threads, each inside an initialised process group
    of the "threaded" backend (default group = all ``world`` ranks).  Returns {rank: "ok" | traceback}."""
    from torch.testing._internal.distributed import multi_threaded_pg as mtpg
    torch._C._distributed_c10d._set_thread_isolation_mode(True)
    world_obj = mtpg._install_threaded_pg()
    store = dist.HashStore()
    res = {}

    def body(rank):
        try:
            dist.init_process_group(backend="threaded", rank=rank, world_size=world, store=store)
            fn(rank, world, *args)
            res[rank] = "ok"
        except BaseException as ex:  # noqa: BLE001
            res[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
            mtpg.ProcessLocalGroup.exception_handle(ex)      # wake the ranks waiting for this one
        finally:
            try:
                if dist.distributed_c10d._world is world_obj:
                    dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass

    import time
    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    deadline = time.monotonic() + timeout          # ONE deadline for all ranks (not world x timeout)
    for t in threads:
        t.join(max(0.0, deadline - time.monotonic()))
    alive = [r for r, t in enumerate(threads) if t.is_alive()]
    for r in alive:
        res.setdefault(r, f"timeout: the rank thread did not finish within {timeout:.0f} s")
    if alive:
        # rank threads are still inside collectives of the threaded process group: it is NOT taken away
        # from under them (they are daemon threads: the caller reports the timeout and ends the process)
        return res
    mtpg._uninstall_threaded_pg()
    torch._C._distributed_c10d._set_thread_isolation_mode(False)
    return res
