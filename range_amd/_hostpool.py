"""Recycling of the host result arrays of the numpy contract.

``model(x)`` hands the caller a host array per call (range/range.py:240).  A fresh 100 MB array
costs first-touch page faults while it is filled and a page-table teardown (munmap) when the caller
drops it - together 8 to 10 ms per 10 000 queries, a third of the GPU time of the batch.  The pool
keeps the memory of results the caller has DROPPED and hands it out again.

Ownership is explicit (round 2 decided it from ``sys.getrefcount`` inside a finalizer - one
interpreter change away from handing live memory out again): the pool owns plain memory blocks
(uint8 arrays nothing else refers to); a
result is ``numpy.frombuffer`` over a per-result GUARD object (a ctypes array created
``from_buffer`` of the block), so the result, every slice / transpose / reshape of it and every
``torch.from_numpy`` tensor over it hold a reference chain to that guard - it is the ``base`` the
views collapse to.  The block goes back on the free list from the guard's finalizer, i.e. only
once the interpreter itself has found the guard unreachable: no reference counts are read, a view
that outlives the result (in another thread, in a reference cycle, under a delayed garbage
collector) keeps the memory out of the pool exactly as long as it lives.  Callers that keep every
result (``save_embeddings`` collects them) never return anything and get fresh memory as before.

What no ownership scheme in Python can see is a consumer holding a RAW POINTER into a result it has
dropped (``arr.ctypes.data`` / ``__array_interface__`` borrowed by a C extension without a
reference to the array).  The reference's arrays are not safe to use that way either (freed memory
instead of recycled memory); ``RANGE_HOST_POOL=0`` in the environment turns the pool off.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref
from typing import Dict, List

import numpy as np


def _size_class(nbytes: int) -> int:
    """Block capacity for a request: the next multiple of 1/8 of the enclosing power of two (at most
    12.5 % over) - a bounded set of capacities, hence of ctypes guard types, whatever batch sizes a
    serving loop sends."""
    if nbytes <= 4096:
        return 4096
    step = 1 << max(12, nbytes.bit_length() - 4)
    return (nbytes + step - 1) // step * step


class HostResultPool:
    #: free blocks kept per capacity, and in total bytes
    max_free_per_size = 2
    max_free_bytes = 1 << 30

    def __init__(self, enabled: bool | None = None):
        self._free: Dict[int, List[np.ndarray]] = {}
        self._free_bytes = 0
        self._lock = threading.Lock()
        self.enabled = (os.environ.get("RANGE_HOST_POOL", "1") != "0") if enabled is None else enabled

    def take(self, rows: int, cols: int) -> np.ndarray:
        """A C-contiguous, writeable float64 array (rows, cols): recycled memory when some is free."""
        nbytes = rows * cols * 8
        if not self.enabled or nbytes == 0:
            return np.empty((rows, cols), dtype=np.float64)
        cap = _size_class(nbytes)
        block = None
        with self._lock:
            lst = self._free.get(cap)
            if lst:
                block = lst.pop()
                self._free_bytes -= cap
        if block is None:
            # untouched pages (bytearray(cap) would memset them: 50 ms per 100 MB in one thread);
            # their first-touch faults are spread over the library's copy threads when the result is filled
            block = np.empty(cap, dtype=np.uint8)
        guard = (ctypes.c_ubyte * cap).from_buffer(block)
        fin = weakref.finalize(guard, self._give_back, block)
        fin.atexit = False
        out = np.frombuffer(guard, dtype=np.float64, count=rows * cols).reshape(rows, cols)
        del guard                               # (only the array and its views hold it now)
        return out

    def _give_back(self, block: np.ndarray) -> None:
        # called by the interpreter once the guard of a result is unreachable: nothing can see the
        # block's memory through numpy any more
        cap = block.nbytes
        with self._lock:
            lst = self._free.setdefault(cap, [])
            if len(lst) < self.max_free_per_size and self._free_bytes + cap <= self.max_free_bytes:
                lst.append(block)
                self._free_bytes += cap


POOL = HostResultPool()
