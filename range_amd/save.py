"""Batch driver: ``save_embeddings`` - the only real caller of ``model(coords)`` in the reference
(range/utils/save.py:7-58; CLI at range/range.py:281-298) - with the device->host copy of the
(B,1280) float64 result taken off the critical path.

The reference does, per batch: H2D of coords, forward, a synchronous D2H of 10 KB per query
(``.cpu()`` at range.py:240), numpy bookkeeping.  Here the batches flow through a two-deep
pipeline: the engine computes batch i on torch's current stream while a copy stream drains batch
i-1 into pinned memory and the host scatters batch i-2 into the final array.  Same signature,
same output files (``np.savez(path, coords=, embeddings=, y=)``), same directory layout.
"""
from __future__ import annotations

import os
from typing import Iterable, Iterator, List, Optional, Tuple

import numpy as np
import torch

from ._hostpool import POOL


class EmbeddingPipeline:
    """Streams batches of (lon,lat) coordinates through a range_amd ``LocationEncoder`` and
    yields host ``numpy`` arrays in order, overlapping compute, D2H and host copies."""

    def __init__(self, model, depth: int = 2):
        self.model = model
        self.engine = model.engine
        self.device = self.engine.device
        self.depth = max(2, int(depth))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        # staging buffers live on the model: page-locking 100 MB takes ~30 ms, which a new
        # pipeline per save_embeddings call (or per loader) would pay again
        cache = model.__dict__.setdefault("_pipeline_staging", {})
        self._pinned, self._dev = cache.setdefault(self.depth, ([None] * self.depth, [None] * self.depth))
        self._done = [torch.cuda.Event() for _ in range(self.depth)]
        self._copied = [torch.cuda.Event() for _ in range(self.depth)]

    def _buffers(self, slot: int, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        dim = self.model.location_feature_dim
        if self._pinned[slot] is None or self._pinned[slot].shape[0] < n:
            self._pinned[slot] = torch.empty((n, dim), dtype=torch.float64).pin_memory()
            self._dev[slot] = torch.empty((n, dim), dtype=torch.float64, device=self.device)
        return self._pinned[slot][:n], self._dev[slot][:n]

    def _coords_async(self, slot: int, coords) -> torch.Tensor:
        """Host coordinates go to the device through a pinned staging buffer with a non-blocking
        copy on the compute stream.  (A plain ``.to(device)`` of pageable memory blocks the host
        until EVERYTHING queued on the device has finished - the previous batch and its 2 ms
        device->host copy - so the next batch was enqueued late and the GPU idled 2-3 ms per batch.)"""
        if torch.is_tensor(coords) and coords.is_cuda:
            return self.model._coords(coords)
        c = coords if torch.is_tensor(coords) else torch.as_tensor(np.asarray(coords))
        if c.dim() != 2 or c.shape[1] != 2:
            raise ValueError(f"coords must be (B,2) (lon,lat) degrees, got {tuple(c.shape)}")
        n = c.shape[0]
        st = self.model.__dict__.setdefault("_pipeline_coords", {})
        key = (self.depth, slot)
        if key not in st or st[key][0].shape[0] < n:
            st[key] = (torch.empty((n, 2), dtype=torch.float64).pin_memory(),
                       torch.empty((n, 2), dtype=torch.float64, device=self.device))
        hp, dv = st[key]
        hp[:n].copy_(c)                       # (also widens float32 input)
        dv[:n].copy_(hp[:n], non_blocking=True)
        return dv[:n]

    @torch.no_grad()
    def run(self, batches: Iterable) -> Iterator[np.ndarray]:
        """``batches`` yields (B,2) coordinate tensors/arrays.  Yields one float64 ndarray
        (B,1280) per batch, in order.  Each yielded array is a fresh host array."""
        if getattr(self.model, "_model_id", None) is None:
            # SatCLIP and the training-free coordinate encoders: tiny outputs, no pipeline -
            # one call per batch like range/utils/save.py:28-30
            for coords in batches:
                out = self.model(coords)
                yield out.cpu().numpy() if torch.is_tensor(out) else np.asarray(out)
            return
        inflight: List[Tuple[int, int]] = []          # (slot, rows)
        compute = torch.cuda.current_stream(self.device)
        i = 0
        for coords in batches:
            slot = i % self.depth
            if len(inflight) == self.depth:           # slot about to be reused: drain it first
                yield self._collect(*inflight.pop(0))
            x = self._coords_async(slot, coords)
            n = x.shape[0]
            pinned, dev = self._buffers(slot, n)
            beta = 1.0 if self.model._model_id == 0 else float(self.model.args.beta)
            for lo in range(0, n, self.model.chunk_size):
                self.engine.forward(x[lo:lo + self.model.chunk_size], self.model._model_id, beta,
                                    out=dev[lo:lo + self.model.chunk_size])
            self._done[slot].record(compute)
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self._done[slot])
                pinned.copy_(dev, non_blocking=True)
                self._copied[slot].record(self.copy_stream)
            inflight.append((slot, n))
            i += 1
        while inflight:
            yield self._collect(*inflight.pop(0))

    def _collect(self, slot: int, n: int) -> np.ndarray:
        self._copied[slot].synchronize()
        self.engine.check_async_error()       # (a persistent launch of this batch that gave up: NaN rows, known now)
        # a fresh array per batch (the reference's contract); its first-touch page faults are
        # spread over the library's copy threads (one thread takes 12 ms per 100 MB)
        src = self._pinned[slot][:n].numpy()
        out = POOL.take(*src.shape)
        self.engine.host_copy(out, src)
        return out


class ShardedEmbeddingPipeline:
    """The batch driver over a ROW-SHARDED model (``load_model(..., shards=W)``): every rank of the
    job iterates the same batches; of a batch of n rows rank r embeds rows [n r / W, n (r+1) / W)
    against all shards (``ShardedRange.embed``: collective inside its shard group) and its (n / W,
    1280) float64 rows travel to rank 0 ONLY - one gather of 10 KB per own query per rank, where
    an all-gather of the full result on every rank moves W times that - on RCCL's stream, while the
    next batch computes; rank 0 copies the gathered batch to pinned memory on a copy stream and
    yields host arrays in order, two batches behind the compute (as ``EmbeddingPipeline``).  The
    other ranks yield nothing."""

    def __init__(self, model, depth: int = 2):
        import torch.distributed as dist
        self.model, self.dist = model, dist
        self.device = model.engine.device
        self.group = model.group
        self.world, self.rank = model.world, model.rank
        self.root_global = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        self.staged = dist.get_backend(self.group) == "gloo"      # (ranks sharing a GPU in tests: through the host)
        self.depth = max(2, int(depth))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._slots = [None] * self.depth       # (send rows, gathered rows on rank 0, pinned copy on rank 0)
        with torch.cuda.device(self.device):
            self._copied = [torch.cuda.Event() for _ in range(self.depth)]
            self._done = [torch.cuda.Event() for _ in range(self.depth)]

    def _buffers(self, slot: int, per: int):
        cur = self._slots[slot]
        if cur is None or cur[0].shape[0] < per:
            send = torch.empty((per, 1280), dtype=torch.float64, device=self.device)
            gathered = pinned = None
            if self.rank == 0:
                gathered = torch.empty((self.world * per, 1280), dtype=torch.float64, device=self.device)
                pinned = torch.empty((self.world * per, 1280), dtype=torch.float64).pin_memory()
            # the ranks' give-up flags of the batch (range_async_error_flag), on EVERY rank: one float64
            # per rank, all-gathered beside the rows and copied to pinned memory with them
            flag = torch.zeros((1,), dtype=torch.float64, device=self.device)
            flags = torch.zeros((self.world,), dtype=torch.float64, device=self.device)
            flags_host = torch.zeros((self.world,), dtype=torch.float64).pin_memory()
            cur = self._slots[slot] = (send, gathered, pinned, flag, flags, flags_host)
        send, gathered, pinned = cur[:3]
        W = self.world
        return send[:per], (None if gathered is None else gathered[:W * per]), (None if pinned is None else pinned[:W * per])

    @torch.no_grad()
    def run(self, batches: Iterable) -> Iterator[np.ndarray]:
        W, r, dist = self.world, self.rank, self.dist
        sharded = self.model.sharded
        inflight: List[Tuple[int, int, int, object]] = []       # (slot, n, per, host rows when staged)
        i = 0
        for coords in batches:
            x = self.model._coords(coords)
            n = x.shape[0]
            if n == 0:
                continue
            slot = i % self.depth
            if len(inflight) == self.depth:
                res = self._collect(*inflight.pop(0))
                if r == 0:
                    yield res
            per = (n + W - 1) // W                                # the largest per-rank count: no collective to agree on it
            lo, hi = (n * r) // W, (n * (r + 1)) // W
            send, gathered, pinned = self._buffers(slot, per)
            # (b_max: the step count of the shard group's collectives; the same on all its ranks)
            sharded.embed(x[lo:hi], out=send[:hi - lo], b_max=per)
            nb = per * 1280 * 8
            sharded._count("results", 0 if r == 0 else nb, (W - 1) * nb if r == 0 else 0)
            flag, flags, flags_host = self._slots[slot][3:]
            if hasattr(self.model.engine, "async_error_flag"):
                self.model.engine.async_error_flag(out=flag)        # (behind this batch's kernels, in stream order)
            host = None
            if self.staged:
                h = send.cpu()
                parts = [torch.empty_like(h) for _ in range(W)] if r == 0 else None
                dist.gather(h, parts, dst=self.root_global, group=self.group)
                fl = torch.zeros((W,), dtype=torch.float64)
                dist.all_gather_into_tensor(fl, flag.cpu(), group=self.group)
                flags_host.copy_(fl)
                host = parts
            else:
                # (recorded on the stream the engine launches on - torch's current stream of the ENGINE's
                # device, whatever device is current in this thread)
                done = self._done[slot]
                done.record(torch.cuda.current_stream(self.device))
                side = self.copy_stream if not sharded.blocking else torch.cuda.current_stream(self.device)
                with torch.cuda.stream(side):
                    # the gather and the device->host copy behind it wait for THIS batch only; the
                    # compute stream goes on with the next batch
                    side.wait_event(done)
                    parts = list(gathered.view(W, per, 1280).unbind(0)) if r == 0 else None
                    dist.gather(send, parts, dst=self.root_global, group=self.group)
                    dist.all_gather_into_tensor(flags, flag, group=self.group)       # (8 B per rank: every rank learns of every give-up)
                    if r == 0:
                        pinned.copy_(gathered, non_blocking=True)
                    flags_host.copy_(flags, non_blocking=True)
                    self._copied[slot].record(side)
            inflight.append((slot, n, per, host))
            i += 1
        while inflight:
            res = self._collect(*inflight.pop(0))
            if r == 0:
                yield res

    def _collect(self, slot: int, n: int, per: int, host):
        W = self.world
        if not self.staged:
            self._copied[slot].synchronize()       # (every rank: its send buffer is free again)
        # A rank whose persistent launch gave up left NaN rows in what it sent and SAID so: its error
        # word travelled with the batch as a flag (range_async_error_flag, all-gathered: 8 B per rank).
        # EVERY rank holds the same flags and refuses the batch here, together - none is left waiting in
        # the next batch's collective - and the verdict comes from the word, not from the data: a NaN
        # or infinite input coordinate gives a NaN row too (as in the reference and in the one-GPU
        # pipeline, whose rows are independent) and passes through like any other row.
        bad = [int(r) for r in np.flatnonzero(self._slots[slot][5].numpy() != 0.0)]
        self.model.engine.check_async_error()      # (this rank's own: reported with the library's message)
        if bad:
            raise RuntimeError(f"sharded save_embeddings: a persistent launch of rank(s) {bad} of the group gave up during this "
                               f"batch of {n} rows (range_hip.h: range_check_async_error; their rows are NaN; each reports it "
                               "on its side and has switched to separate launches) - re-run the batch")
        if self.rank != 0:
            return None
        out = POOL.take(n, 1280)
        for r in range(W):
            lo, hi = (n * r) // W, (n * (r + 1)) // W
            src = host[r][:hi - lo].numpy() if host is not None else self._slots[slot][2][r * per:r * per + hi - lo].numpy()
            out[lo:hi] = src
        return out


def save_embeddings(args, train_loader, val_loader, location_model):
    """Drop-in for range/utils/save.py:7-58."""
    embeddings_dir = os.path.join(args.embeddings_dir, args.location_model_name)
    if not os.path.exists(embeddings_dir):
        print(f"Creating new directory {embeddings_dir}")
        os.makedirs(embeddings_dir, exist_ok=True)       # (several ranks of a sharded job may get here together)
    train_path = os.path.join(embeddings_dir, f"{args.task_name}_train.npz")
    val_path = os.path.join(embeddings_dir, f"{args.task_name}_val.npz")
    location_model.eval()
    if getattr(location_model, "is_sharded", False):
        # row-sharded model (load_model(..., shards=W)): every rank of the job runs this function over
        # the SAME loaders and embeds ITS rows of every batch; the rows travel to rank 0 only, which
        # writes the files (ShardedEmbeddingPipeline)
        import torch.distributed as dist
        pipe = ShardedEmbeddingPipeline(location_model)
        root = dist.get_rank(location_model.group) == 0
        for loader, path in ((train_loader, train_path), (val_loader, val_path)):
            coords_list, y_list = [], []

            def coords_iter():
                for coords, y in loader:                                 # range/utils/save.py:24-37
                    if root:
                        coords_list.append(coords.cpu().numpy() if torch.is_tensor(coords) else np.asarray(coords))
                        y_list.append(y.cpu().numpy() if torch.is_tensor(y) else np.asarray(y))
                    yield coords

            embeddings_list = list(pipe.run(coords_iter()))
            if root:
                np.savez(path, coords=np.concatenate(coords_list, axis=0),
                         embeddings=np.concatenate(embeddings_list, axis=0),
                         y=np.concatenate(y_list, axis=0))
                print(f"File saved to {path}")
        dist.barrier(location_model.group)
        if root:
            print(f"File saved to {train_path} and {val_path}")
        return
    pipe = EmbeddingPipeline(location_model)
    for loader, path in ((train_loader, train_path), (val_loader, val_path)):
        coords_list, y_list = [], []

        def coords_iter():
            for coords, y in loader:
                coords_list.append(coords.cpu().numpy() if torch.is_tensor(coords) else np.asarray(coords))
                y_list.append(y.cpu().numpy() if torch.is_tensor(y) else np.asarray(y))
                yield coords

        embeddings_list = list(pipe.run(coords_iter()))
        np.savez(path, coords=np.concatenate(coords_list, axis=0),
                 embeddings=np.concatenate(embeddings_list, axis=0),
                 y=np.concatenate(y_list, axis=0))
        print(f"File saved to {path}")
    print(f"File saved to {train_path} and {val_path}")
