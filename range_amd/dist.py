"""Row-sharded bank across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI).

The reference has no distributed code; this is the north-star layout: rank r
holds bank rows [r*N/W, (r+1)*N/W) and serves its own B queries; the soft-attention over N
decomposes exactly over row shards:

    1. encode the local queries (kernel A)                       no communication
    2. all-gather of the query operands e32 (B,256) and xq (B,4): W*B*1040 B per rank
    3. pass 1 on the local shard for ALL W*B queries -> softmax statistics (m, l) with the
       constant shift m = tau*log2(e) (unit-vector logits), so the l of disjoint shards ADD
    4. all-gather of the statistics (W*B,4) + merge in rank order (same m, the l add: a fixed
       order, where an all-reduce's would depend on an element's place in the buffer)
    5. pass 2 on the local shard with the GLOBAL statistics -> partial (W*B,1024) f32; partials
       of different shards simply add because the weights are already globally normalised
    6. all-to-all: rank r receives the W partial slices of ITS queries (one direct transfer per
       peer, so all 7 xGMI links of a GPU carry 1/7 of the traffic each - not a ring)
    7. finalize: fixed-order sum of the W slices + pack with e64 -> (B,1280) float64

Steps 2-7 run per query CHUNK and every collective is asynchronous (RCCL's own stream): the
gathers of all chunks are issued behind the encoder, pass 1 of chunk c runs while the gather of
chunk c+1 and the statistics of chunk c-1 travel, pass 2 of chunk c while the exchange of chunk c-1
travels.  Exposed on the compute stream: the first chunk's gather and the last chunk's exchange.
Pass 1 keeps the logits of all chunks in ONE workspace (range_scan_stats_at), every chunk with the
same bank-split count, so a query's statistics - and the result - do not depend on the chunking.

The top-k side channel merges per-shard candidate lists with ONE all-gather (north star).

Per-GPU work is B_total * N / W = B * N: adding GPUs adds queries at constant time per step
(weak scaling).  The engine is duck-typed (see ``_native.HipEngine``) so that the collective
logic can be exercised on CPU with the gloo backend and a checker engine in tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from .range import TEMP_GEO, TEMP_RANGE, TEMP_RANGE_PLUS


def shard_rows(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range of ``rank``."""
    return (n_rows * rank) // world_size, (n_rows * (rank + 1)) // world_size


class ShardedRange:
    """RANGE / RANGE+ forward over a row-sharded bank.  ``engine`` holds THIS rank's rows."""

    #: queries per rank and chunk below which a forward is not split further.  Chunks buy overlap -
    #: the statistics of chunk c travel under pass 1 of chunk c+1, its exchange under pass 2 of chunk
    #: c+1, so only the first gather and the last exchange are exposed - and cost launches: what a rank
    #: of 8 computes for BASELINE's 10 000-query batch (1 250 own, 10 000 scanned against 12 500 rows;
    #: tools/shard_emulate.py --chunks k 8, round 4) takes 2.885 ms as one chunk, 3.011 ms as two,
    #: 3.119 ms as four.  Default: TWO chunks from 1 024 queries per rank (three blocking
    #: collectives -> one gather + half an exchange exposed), four from 8 192 (the weak mode, where a
    #: rank ships >= 8 MB per peer and chunk)
    min_chunk = 512

    def __init__(self, engine, model_name: str = "RANGE+", beta: Optional[float] = 0.5,
                 group=None, n_chunks: Optional[int] = None):
        if model_name == "RANGE":
            self.tau_sem, self.tau_geo, self.beta = TEMP_RANGE, 0.0, 1.0
        elif model_name == "RANGE+":
            self.tau_sem, self.tau_geo, self.beta = TEMP_RANGE_PLUS, TEMP_GEO, float(beta)
        else:
            raise ValueError("Unimplemented RANGE model")
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_chunks = n_chunks   # None: 2 or 4 chunks (by batch size) when there is a peer to exchange with
        # gathered / chunk-major / receive buffers: one per name, sized to the largest request seen
        # (a step then makes no allocation of its own besides the engine's outputs)
        self._bufs = {}
        self._timing = False
        self._events = {}
        #: pass 1 per chunk with its collectives overlapped (needs engine.scan_stats_at); False: one
        #: pass 1 over all scanned queries between a blocking gather and a blocking all-reduce (A/B, tests)
        self.pass1_chunked = hasattr(engine, "scan_stats_at")
        #: RANGE_DIST_BLOCKING=1: every collective is waited for where it is issued (no overlap with
        #: compute, one pass 1 over all scanned queries) - the escape hatch for bisecting a hang or a
        #: wrong result on a backend the overlapped schedule has not met (it has run over gloo and
        #: the in-process test backend, and over RCCL with ONE rank; with more than one rank it has not executed)
        self.blocking = os.environ.get("RANGE_DIST_BLOCKING", "0") == "1"
        if self.blocking:
            self.pass1_chunked = False
        #: bytes this rank SENT / RECEIVED per kind of collective since ``reset_bytes()`` (payload
        #: sizes; "results" = what the batch drivers move to rank 0, range.ShardedLocationEncoder)
        self.bytes_sent = {}
        self.bytes_received = {}

    def reset_bytes(self):
        self.bytes_sent, self.bytes_received = {}, {}

    def _count(self, kind: str, sent: int, received: int):
        self.bytes_sent[kind] = self.bytes_sent.get(kind, 0) + int(sent)
        self.bytes_received[kind] = self.bytes_received.get(kind, 0) + int(received)

    def _buf(self, name: str, shape, dtype, device) -> torch.Tensor:
        """One flat buffer per (name, dtype, device), grown to the largest request seen; a request is
        a view of its head.  (A buffer per distinct shape would grow without bound under varying
        batch sizes: ragged last batches, a serving loop.)"""
        key = (name, dtype, str(device))
        n = 1
        for d in shape:
            n *= int(d)
        t = self._bufs.get(key)
        if t is None or t.numel() < n:
            t = self._bufs[key] = torch.empty(max(n, 1), dtype=dtype, device=device)
        return t[:n].view(tuple(shape))

    # -- exposed communication time ------------------------------------------------------------
    def comm_timing(self, on: bool):
        """``comm_timing(True)`` starts measuring, ``comm_timing(False)`` stops and returns the
        milliseconds the COMPUTE stream spent blocked on collectives since, per kind - a dict
        {"gather", "reduce", "exchange", "total"} (event pairs around every point where the stream
        waits for one: the time a collective - including the wait for the slowest peer - is not
        hidden behind this rank's kernels).  None without a GPU."""
        if on:
            self._events, self._timing = {}, torch.cuda.is_available()
            return None
        if not self._timing:
            return None
        self._timing = False
        torch.cuda.synchronize()
        ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self._events.items()}
        for k in ("gather", "reduce", "exchange"):
            ms.setdefault(k, 0.0)
        ms["total"] = sum(ms.values())
        self._events = {}
        return ms

    def _blocked(self, kind: str, fn):
        """Run ``fn`` (something that makes the current stream wait for a collective) between two
        events on the current stream."""
        if not self._timing:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self._events.setdefault(kind, []).append((a, b))
        return r

    def _staged(self, t: torch.Tensor) -> bool:
        """Device tensors over the gloo backend (several ranks sharing one GPU in tests, or a box
        without RCCL): the collective runs on host copies.  RCCL moves device memory directly."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _gather(self, t: torch.Tensor, name: str) -> torch.Tensor:
        # concatenation form (W*n, ...): accepted by both the RCCL and the gloo backend
        src = t.contiguous()
        if self._staged(t):
            src = src.cpu()
        out = self._buf("gather:" + name, (self.world * t.shape[0],) + tuple(t.shape[1:]), t.dtype,
                        src.device)
        self._blocked("gather", lambda: dist.all_gather_into_tensor(out, src, group=self.group))
        nb = src.numel() * src.element_size()
        self._count("gather", nb * (self.world - 1), nb * (self.world - 1))
        return out.to(t.device).reshape((self.world,) + tuple(t.shape))

    def _gather_start(self, src: torch.Tensor, dst: torch.Tensor):
        """Start the all-gather of ``src`` (n, d) into ``dst`` (W*n, d), rank-major: returns a
        function that makes the current stream wait for it (and, staged, lands the host copy)."""
        nb = src.numel() * src.element_size()
        self._count("gather", nb * (self.world - 1), nb * (self.world - 1))
        if self._staged(src):
            h_src = src.cpu()
            h_dst = torch.empty(dst.shape, dtype=dst.dtype)
            work = dist.all_gather_into_tensor(h_dst, h_src, group=self.group, async_op=True)

            def land():
                work.wait()
                dst.copy_(h_dst)
            return land
        work = dist.all_gather_into_tensor(dst, src.contiguous(), group=self.group, async_op=True)
        if self.blocking:
            work.wait()
            return lambda: None
        return work.wait

    def _stats_start(self, stats: torch.Tensor, name: str):
        """Start the all-gather of a chunk's statistics (n,4) and return the function that waits for
        it and merges the W shards' rows in rank order (engine.merge_stats: same shift m, the l add).
        Not an all-reduce: a ring or tree sums an element in an order that depends on where it sits
        in the buffer, so a query's l - and with it the result - would change in the last bit with
        the batch it travels in; the fixed-order merge keeps the sharded result independent of the
        chunking and of the collective algorithm (16 B per scanned query to every peer, one small
        kernel).  Counted and timed under the key "reduce" (what it replaces); the time is the
        wait for the collective alone, not the merge kernel behind it."""
        out = self._buf("gather:" + name, (self.world * stats.shape[0], stats.shape[1]), stats.dtype,
                        torch.device("cpu") if self._staged(stats) else stats.device)
        nb = stats.numel() * stats.element_size()
        self._count("reduce", nb * (self.world - 1), nb * (self.world - 1))
        src = stats.cpu() if self._staged(stats) else stats
        work = dist.all_gather_into_tensor(out, src, group=self.group, async_op=True)
        if self.blocking:
            self._blocked("reduce", work.wait)

        def merged():
            if not self.blocking:
                self._blocked("reduce", work.wait)
            parts = out.to(stats.device).view(self.world, stats.shape[0], stats.shape[1])
            return self.engine.merge_stats(parts)
        return merged

    def _all_to_all(self, part: torch.Tensor, name: str):
        """Start the exchange of a chunk's partials; returns (work, getter of the received tensor)."""
        nb = part.numel() * part.element_size() // self.world * (self.world - 1)
        self._count("exchange", nb, nb)
        if self._staged(part):
            src = part.cpu()
            recv = torch.empty_like(src)
            work = dist.all_to_all_single(recv, src, group=self.group, async_op=True)
            return work, (lambda: recv.to(part.device)), src
        recv = self._buf("recv:" + name, part.shape, part.dtype, part.device)
        # one direct transfer per peer; asynchronous, so that the exchange of this chunk
        # overlaps pass 2 of the next
        work = dist.all_to_all_single(recv, part, group=self.group, async_op=True)
        if self.blocking:
            self._blocked("exchange", work.wait)
        return work, (lambda: recv), part

    def _reduce_stats(self, stats_local: torch.Tensor) -> torch.Tensor:
        """Global softmax statistics from the shards' (the blocking form of the unchunked pass 1):
        every shard reports (m, l) with the SAME constant shift m (range_hip.h: range_scan_stats),
        so the sums l of disjoint row sets add - gathered and merged in rank order (``_stats_start``)."""
        return self._stats_start(stats_local, "stats")()

    def _gather_queries(self, lonlat: torch.Tensor):
        """Encode the own queries and gather every rank's scan operands: e32 (B,256) and xq (B,4)
        travel as ONE packed (B,260) buffer - at these sizes a collective costs its latency, not
        its bytes.  Returns e64 (own) and the (W*B,256) / (W*B,4) column views of the gathered
        buffer (strided: ``_chunk_major`` lays them out for the scan)."""
        e64, e32, xq = self.engine.encode(lonlat)
        W, B, d = self.world, lonlat.shape[0], e32.shape[1]
        packed = self._buf("pack:q", (B, d + xq.shape[1]), e32.dtype, e32.device)
        packed[:, :d] = e32
        packed[:, d:] = xq
        allq = self._gather(packed, "q").reshape(W * B, d + xq.shape[1])
        return e64, allq[:, :d], allq[:, d:]

    def _chunk_major(self, t_all: torch.Tensor, chunks, name: str) -> torch.Tensor:
        """(W*B, d) in rank-major order (a column view of the gathered buffer) -> contiguous,
        chunk-major order: rows [lo,hi) of EVERY rank's queries form one chunk, ordered by owner
        rank - a contiguous range of the scanned batch whose partial is again W equal slices, one
        per destination."""
        out = self._buf("cm:" + name, t_all.shape, t_all.dtype, t_all.device)
        if len(chunks) == 1:
            out.copy_(t_all)
            return out
        W = self.world
        B = t_all.shape[0] // W
        v = t_all.view(W, B, -1)
        for lo, hi in chunks:
            out[W * lo:W * hi].view(W, hi - lo, -1).copy_(v[:, lo:hi])
        return out

    def _chunk_bounds(self, B: int):
        """Row ranges [lo,hi) of a rank's queries per chunk.  Boundaries are multiples of 64 (the
        query-tile size: a chunk then starts on a tile of the kept logits)."""
        W = self.world
        n_chunks = self.n_chunks if self.n_chunks else ((4 if B >= 8192 else 2) if W > 1 else 1)
        n_chunks = max(1, min(n_chunks, B // self.min_chunk))
        cuts = sorted({min(B, ((B * c) // n_chunks + 32) // 64 * 64) for c in range(1, n_chunks)})
        bounds = [0] + [c for c in cuts if 0 < c < B] + [B]
        return list(zip(bounds[:-1], bounds[1:]))

    def _scan(self, lonlat: torch.Tensor):
        """Steps 1-4 for this rank's (B,2) queries.  Returns (e64 of the own queries, e32 and xq of
        ALL W*B scanned queries in chunk-major order, the chunk bounds, one function per chunk that
        returns the chunk's GLOBAL statistics once the current stream may read them, and whether pass 2
        finds the scan's logits kept)."""
        W, B = self.world, lonlat.shape[0]
        eng = self.engine
        chunks = self._chunk_bounds(B)
        if not self.pass1_chunked:
            e64, e32_all, xq_all = self._gather_queries(lonlat)
            e32_all = self._chunk_major(e32_all, chunks, "e32")
            xq_all = self._chunk_major(xq_all, chunks, "xq")
            n_max = W * max(hi - lo for lo, hi in chunks)
            if hasattr(eng, "scan_stats_at"):
                # (the same bank splits as the chunked form: bit-identical statistics)
                stats_local = eng.scan_stats_at(e32_all, xq_all, self.tau_sem, self.tau_geo, 0, W * B,
                                                n_splits=eng.p1_splits(n_max))
            else:
                # pass 1 on the local shard keeps its logits; pass 2 reads them back instead of
                # recomputing e . K^T (they are independent of the global statistics)
                stats_local = eng.scan_stats(e32_all, xq_all, self.tau_sem, self.tau_geo, keep_logits=True)
            stats = self._reduce_stats(stats_local)
            getters = [(lambda lo=lo, hi=hi: stats[W * lo:W * hi]) for lo, hi in chunks]
            return e64, e32_all, xq_all, chunks, getters, eng.kept_queries() == W * B
        e64, e32, xq = eng.encode(lonlat)
        total = W * B
        e32_all = self._buf("cm:e32", (total, e32.shape[1]), e32.dtype, e32.device)
        xq_all = self._buf("cm:xq", (total, xq.shape[1]), xq.dtype, xq.device)
        # every chunk's gather is issued now: a chunk's rows of all ranks land rank-major in rows
        # [W lo, W hi) - the chunk-major order of the scan, no copy
        landed = [(self._gather_start(e32[lo:hi], e32_all[W * lo:W * hi]),
                   self._gather_start(xq[lo:hi], xq_all[W * lo:W * hi])) for lo, hi in chunks]
        n_splits = eng.p1_splits(W * max(hi - lo for lo, hi in chunks))
        getters = []
        for (lo, hi), (we, wx) in zip(chunks, landed):
            self._blocked("gather", we)
            self._blocked("gather", wx)
            first, n = W * lo, W * (hi - lo)
            st = eng.scan_stats_at(e32_all[first:first + n], xq_all[first:first + n], self.tau_sem, self.tau_geo,
                                   first, total, n_splits=n_splits)
            merged = self._stats_start(st, f"stats{len(getters)}")
            getters.append(merged)
        return e64, e32_all, xq_all, chunks, getters, eng.kept_queries() == total

    def _topk_start(self, e32_all: torch.Tensor, chunks, k: int):
        """The top-k side channel of a forward (``forward(..., topk=k)``): this shard's top-k for ALL
        scanned queries - their e-hat is gathered already, in chunk-major order - and the start of
        the ONE all-gather of the packed candidates (north star); returns the function that waits for
        it and merges this rank's OWN queries' W lists (global rows, ties to the lower row)."""
        W, total = self.world, e32_all.shape[0]
        B = total // W
        # (the keys-only scan: it leaves the logits pass 1 kept for pass 2 alone)
        tv, ti = self.engine.topk_stream(e32_all, k)
        packed = self._buf("pack:topk", (total, k, 3), torch.float32, tv.device)
        packed[:, :, 0] = tv
        packed[:, :, 1:] = ti.view(torch.float32).reshape(total, k, 2)          # int64 bit pattern
        staged = self._staged(packed)
        src = packed.cpu() if staged else packed
        allp = self._buf("gather:fwd_topk", (W * total, k, 3), torch.float32, src.device)
        nb = src.numel() * src.element_size()
        self._count("topk", nb * (W - 1), nb * (W - 1))
        work = dist.all_gather_into_tensor(allp, src, group=self.group, async_op=True)
        if self.blocking:
            self._blocked("gather", work.wait)

        def merged():
            if not self.blocking:
                self._blocked("gather", work.wait)
            a = allp.to(tv.device).view(W, total, k, 3)
            # this rank's rows of the chunk-major order: chunk (lo,hi) holds them at W lo + rank (hi - lo)
            own = torch.cat([a[:, W * lo + self.rank * (hi - lo):W * lo + (self.rank + 1) * (hi - lo)] for lo, hi in chunks], dim=1)
            vals = own[..., 0].contiguous()
            idxs = own[..., 1:].contiguous().view(torch.int64).reshape(W, B, k)
            return self.engine.merge_topk(vals, idxs)
        return merged

    @torch.no_grad()
    def forward(self, lonlat: torch.Tensor, out: Optional[torch.Tensor] = None, topk: Optional[int] = None):
        """lonlat: this rank's (B,2) float64 queries (same B on every rank).
        Returns this rank's (B,1280) float64 embeddings (device tensor; ``out`` when given: the
        chunks are finalized straight into its rows).  ``topk=k``: returns (embeddings, values (B,k)
        float32, global bank rows (B,k) int64) - the top-k side channel from the same call: the
        queries are encoded and gathered ONCE, every shard scans the gathered e-hat behind its pass 1,
        the candidates travel in one all-gather under pass 2."""
        W = self.world
        e64, e32_all, xq_all, chunks, stats_of, kept = self._scan(lonlat)
        topk_of = self._topk_start(e32_all, chunks, int(topk)) if topk else None
        pending = []
        for ci, (lo, hi) in enumerate(chunks):
            first, n = W * lo, W * (hi - lo)
            stats = stats_of[ci]()
            if kept:
                part = self.engine.attend_kept(first, xq_all[first:first + n], self.tau_sem,
                                               self.tau_geo, self.beta, stats)
            else:
                part = self.engine.attend(e32_all[first:first + n], xq_all[first:first + n],
                                          self.tau_sem, self.tau_geo, self.beta, stats)
            work, get, keep = self._all_to_all(part, f"fwd{ci}")
            pending.append((work, get, keep, lo, hi))
        if out is None:
            out = torch.empty((lonlat.shape[0], e64.shape[1] + 1024), dtype=torch.float64, device=e64.device)
        into = hasattr(self.engine, "scan_stats_at")      # (engines of the round-3 duck type return a new tensor)
        for work, get, keep, lo, hi in pending:
            self._blocked("exchange", work.wait)
            recv = get().reshape(W, hi - lo, -1)
            if into:
                self.engine.finalize(recv, e64[lo:hi], out=out[lo:hi])
            else:
                out[lo:hi] = self.engine.finalize(recv, e64[lo:hi].contiguous())
        if topk_of is not None:
            tv, ti = topk_of()
            return out, tv, ti
        return out

    __call__ = forward

    #: scanned queries (all ranks') per outer step of ``embed`` / ``embed_sweep``: bounds the
    #: per-rank workspace like LocationEncoder.chunk_size does on one GPU (kept logits: 4 B x
    #: scanned queries x local rows; split slabs: splits x scanned queries x 4 KB)
    scan_chunk = 16384

    def _max_over_ranks(self, n: int, device) -> int:
        t = torch.tensor([n], dtype=torch.int64,
                         device=device if dist.get_backend(self.group) != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _steps(self, lonlat: torch.Tensor, chunk: Optional[int], b_max: Optional[int] = None):
        """Outer steps of a ragged job: every rank brings its OWN number of queries (0 allowed); the
        ranks agree on the largest count (one scalar all-reduce), walk it in steps of ``chunk``
        queries per rank and pad their share of a step to the step's size with copies of a dummy
        location - every collective of a step then has the same shape on every rank.  Yields
        (lo, n_own, padded (c,2) tensor)."""
        if lonlat.dim() != 2 or lonlat.shape[1] != 2:
            raise ValueError(f"lonlat must be (B,2) (lon,lat) degrees, got {tuple(lonlat.shape)}")
        B = lonlat.shape[0]
        if b_max is None:      # (a caller that knows every rank brings the same count passes it: no collective)
            b_max = self._max_over_ranks(B, lonlat.device)
        elif B > b_max:
            # (rows beyond b_max would never be computed; b_max must be the SAME on every rank of the
            # group - the steps' collectives are shaped by it - and cover every rank's count)
            raise ValueError(f"this rank brings {B} queries but b_max={b_max}")
        if chunk is None:
            chunk = max(64, self.scan_chunk // self.world // 64 * 64)
        chunk = max(1, int(chunk))
        for lo in range(0, b_max, chunk):
            c = min(chunk, b_max - lo)
            own = lonlat[lo:lo + c]
            n_own = own.shape[0]
            if n_own < c:
                pad = self._buf("pad:q", (c, 2), lonlat.dtype, lonlat.device)
                pad.zero_()
                pad[:n_own] = own
                own = pad
            yield lo, n_own, own.contiguous()

    @torch.no_grad()
    def embed(self, lonlat: torch.Tensor, chunk: Optional[int] = None,
              out: Optional[torch.Tensor] = None, b_max: Optional[int] = None, topk: Optional[int] = None):
        """The product entry of the row-sharded path: this rank's (B,2) queries -> its (B,1280)
        float64 embeddings (device tensor), for ANY per-rank B (ragged across ranks, zero on some),
        in outer steps of ``chunk`` queries per rank (default: ``scan_chunk`` scanned queries per
        step).  Collective: every rank of the group must call it.  ``b_max``: the largest per-rank
        count, when the caller knows it (equal counts: ``b_max=B``) - saves the scalar all-reduce.
        ``topk=k``: returns (embeddings, values (B,k), global rows (B,k)) - see ``forward``."""
        B = lonlat.shape[0]
        if out is None:
            out = torch.empty((B, 1280), dtype=torch.float64, device=lonlat.device)
        if topk:
            tv = torch.empty((B, topk), dtype=torch.float32, device=lonlat.device)
            ti = torch.empty((B, topk), dtype=torch.int64, device=lonlat.device)
        for lo, n_own, q in self._steps(lonlat, chunk, b_max):
            if topk:
                res, v, i = self.forward(q, topk=topk)
                if n_own:
                    out[lo:lo + n_own], tv[lo:lo + n_own], ti[lo:lo + n_own] = res[:n_own], v[:n_own], i[:n_own]
                continue
            if n_own == q.shape[0]:
                self.forward(q, out=out[lo:lo + n_own])      # (no padding: straight into the result)
                continue
            res = self.forward(q)
            if n_own:
                out[lo:lo + n_own] = res[:n_own]
        return (out, tv, ti) if topk else out

    @torch.no_grad()
    def embed_sweep(self, lonlat: torch.Tensor, betas, chunk: Optional[int] = None,
                    out: Optional[torch.Tensor] = None, b_max: Optional[int] = None) -> torch.Tensor:
        """``embed`` for several beta values at once (one pass 1 and two passes 2 per step whatever
        the number of betas): (len(betas), B, 1280) float64 on the device."""
        betas = [float(b) for b in betas]
        B = lonlat.shape[0]
        if out is None:
            out = torch.empty((len(betas), B, 1280), dtype=torch.float64, device=lonlat.device)
        for lo, n_own, q in self._steps(lonlat, chunk, b_max):
            res = self.sweep(q, betas)
            if n_own:
                out[:, lo:lo + n_own] = res[:, :n_own]
        return out

    @torch.no_grad()
    def embed_topk(self, lonlat: torch.Tensor, k: int = 16, chunk: Optional[int] = None):
        """``topk`` for ragged per-rank batches: (values (B,k) f32, global indices (B,k) i64)."""
        B = lonlat.shape[0]
        tv = torch.empty((B, k), dtype=torch.float32, device=lonlat.device)
        ti = torch.empty((B, k), dtype=torch.int64, device=lonlat.device)
        for lo, n_own, q in self._steps(lonlat, chunk):
            v, i = self.topk(q, k)
            if n_own:
                tv[lo:lo + n_own] = v[:n_own]
                ti[lo:lo + n_own] = i[:n_own]
        return tv, ti

    @torch.no_grad()
    def sweep(self, lonlat: torch.Tensor, betas) -> torch.Tensor:
        """RANGE+ embeddings of this rank's queries for several beta values (BASELINE config
        "beta sweep ... 8xMI355X"): one pass 1, per chunk TWO passes 2 on its kept logits (beta = 1:
        the semantic retrieval H, beta = 0: the geographic G), two exchanges, then one blend
        (range.py:238, applied per shard partial - the blend is linear) + finalize per beta.
        Returns (len(betas), B, 1280) float64 on the device."""
        if self.tau_geo <= 0.0:
            raise ValueError("sweep() is defined for RANGE+ only")
        betas = [float(b) for b in betas]
        W, B = self.world, lonlat.shape[0]
        e64, e32_all, xq_all, chunks, stats_of, kept = self._scan(lonlat)
        pending = []
        for ci, (lo, hi) in enumerate(chunks):
            first, n = W * lo, W * (hi - lo)
            sl = slice(first, first + n)
            st = stats_of[ci]()
            parts = []
            for b in (1.0, 0.0):
                if kept:
                    parts.append(self.engine.attend_kept(first, xq_all[sl], self.tau_sem,
                                                         self.tau_geo, b, st))
                else:
                    parts.append(self.engine.attend(e32_all[sl], xq_all[sl], self.tau_sem,
                                                    self.tau_geo, b, st))
            ex = [self._all_to_all(p, f"sweep{ci}:{j}") for j, p in enumerate(parts)]
            pending.append((ex, lo, hi))
        out = torch.empty((len(betas), B, e64.shape[1] + 1024), dtype=torch.float64,
                          device=e64.device)
        for ex, lo, hi in pending:
            for work, _, _ in ex:
                self._blocked("exchange", work.wait)
            rH, rG = ex[0][1](), ex[1][1]()
            e = e64[lo:hi].contiguous()
            for j, b in enumerate(betas):
                mix = self.engine.blend(rG, rH, b)                   # (W*n, 1024): per-shard partials
                out[j, lo:hi] = self.engine.finalize(mix.reshape(W, hi - lo, mix.shape[1]), e)
        return out

    @torch.no_grad()
    def topk(self, lonlat: torch.Tensor, k: int = 16):
        """Global top-k (semantic cosine similarity) for this rank's queries: per-shard top-k,
        one all-gather of the candidates, k-way merge."""
        W, B = self.world, lonlat.shape[0]
        _, e32_all, xq_all = self._gather_queries(lonlat)
        e32_all, xq_all = e32_all.contiguous(), xq_all.contiguous()
        if hasattr(self.engine, "topk_stream"):
            # the HBM-streaming kernel: faster than pass 1's own top-k at every batch size
            tv, ti = self.engine.topk_stream(e32_all, k)
        else:
            _, tv, ti = self.engine.scan_stats(e32_all, xq_all, self.tau_sem, 0.0, topk=k)
        # values and int64 indices travel in one buffer (index bit patterns viewed as 2 x f32)
        packed = torch.empty((W * B, k, 3), dtype=torch.float32, device=tv.device)
        packed[:, :, 0] = tv
        packed[:, :, 1:] = ti.view(torch.float32).reshape(W * B, k, 2)   # int64 bit pattern
        allp = self._gather(packed, "topk")                                    # the ONE all-gather
        sl = slice(self.rank * B, (self.rank + 1) * B)
        vals = allp[:, sl, :, 0].contiguous()
        idxs = allp[:, sl, :, 1:].contiguous().view(torch.int64).reshape(W, B, k)
        return self.engine.merge_topk(vals, idxs)


def make_layout(row_shards: int, group=None):
    """The 2-D layout R x Q of a W-rank job (W = R Q): the bank is row-sharded over R ranks - a SHARD
    GROUP: ranks g R .. g R + R - 1 - and there are Q such groups, each a full copy of the bank serving
    its own queries.  R = W is the north-star layout (one group, every collective spans the node);
    R = 1 is the query-sharded control (no collective on the data path); in between the collectives
    span R ranks only, the all-to-all is (R - 1) transfers per rank, a shard has N / R rows (pass 2
    keeps its whole-round efficiency on bigger shards) and a rank scans R times its own queries
    instead of W times.  With the default group (``group=None``) every rank of the job must call
    this: the sub-groups are created collectively, each rank entering every ``new_group``.  With a
    sub-group of the job as ``group`` only ITS ranks need to call (they cannot make the other
    ranks enter ``new_group``): each creates just its own shard group with torch's local
    synchronisation.  Returns (shard_group, shard_index, query_group_index)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    R = int(row_shards)
    if R < 1 or world % R:
        raise ValueError(f"row_shards={row_shards} does not divide the {world} ranks")
    if R == world:
        return group, rank, 0
    if group is None or group is dist.group.WORLD:
        ranks = list(range(world))
        mine = None
        for g in range(world // R):
            sub = dist.new_group(ranks=ranks[g * R:(g + 1) * R])
            if rank // R == g:
                mine = sub
        return mine, rank % R, rank // R
    ranks = dist.get_process_group_ranks(group)
    g = rank // R
    try:
        mine = dist.new_group(ranks=ranks[g * R:(g + 1) * R], use_local_synchronization=True)
    except TypeError as ex:       # (a torch without local synchronisation: every rank of the JOB would have to call)
        raise RuntimeError("row_shards < group size over a non-default process group needs a torch whose "
                           "new_group() takes use_local_synchronization") from ex
    return mine, rank % R, rank // R


#: seconds a collective (and the rendezvous) may take before the process group gives up - torch's
#: default is 600 s, the whole budget of a benchmark run: a rank that never arrives must cost two
#: minutes, not the job.  RANGE_DIST_TIMEOUT_S overrides.  Over RCCL the watchdog thread of
#: ProcessGroupNCCL aborts the communicator and ends the process when a collective exceeds it
#: (TORCH_NCCL_ASYNC_ERROR_HANDLING's default); over gloo the waiting call raises.  Either way every
#: rank of a job whose peer hangs exits non-zero within this bound.
DEFAULT_TIMEOUT_S = 120.0


def dist_timeout_s(timeout_s: Optional[float] = None) -> float:
    if timeout_s is not None:
        return float(timeout_s)
    return float(os.environ.get("RANGE_DIST_TIMEOUT_S", DEFAULT_TIMEOUT_S))


def init_from_env(backend: Optional[str] = None, timeout_s: Optional[float] = None,
                  attempt: Optional[int] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).

    ``timeout_s``: the process group's timeout (default ``DEFAULT_TIMEOUT_S`` = 120 s, or
    RANGE_DIST_TIMEOUT_S).  ``attempt`` (default: RANGE_DIST_ATTEMPT, else none): a job whose rank
    processes are started a second time against the SAME store (tools/rank_guard.py: fresh children
    after a failed first contact; under torchrun the store lives in the agent and survives them)
    rendezvouses under a prefix of its own, so that nothing the first attempt left in the store -
    rank addresses, RCCL's unique id - is read by the second."""
    from datetime import timedelta

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if attempt is None and os.environ.get("RANGE_DIST_ATTEMPT"):
        attempt = int(os.environ["RANGE_DIST_ATTEMPT"])
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        timeout = timedelta(seconds=dist_timeout_s(timeout_s))
        kw = dict(rank=rank, world_size=world, timeout=timeout)
        if attempt is not None and attempt > 1:
            store, _, _ = next(dist.rendezvous("env://", rank=rank, world_size=world, timeout=timeout))
            store.set_timeout(timeout)
            kw["store"] = dist.PrefixStore(f"range_attempt{attempt}", store)
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, local, world
