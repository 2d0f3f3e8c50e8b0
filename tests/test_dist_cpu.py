"""world_size-2 gloo tests (CPU) of the row-sharded path's collective logic (range_amd/dist.py).

No GPU here, so the per-shard compute is provided by a CHECKER engine built on the CPU oracle
(allowed in tests only); what is under test is the product's sharding / merge / exchange code:
all-gather of query operands, exact log-sum-exp merge of shard statistics, all-to-all of the
partials and the fixed-order finalize, and the single-all-gather top-k merge."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import range_oracle as O
from tools import synth
from range_amd.dist import ShardedRange, shard_rows

LOG2E = 1.4426950408889634


class OracleShardEngine:
    """Duck-type of _native.HipEngine over one bank shard, float64 math from the oracle."""

    def __init__(self, weights, L, bank: O.Bank, row_offset: int):
        self.w, self.L, self.bank, self.row_offset = weights, L, bank, row_offset
        self._kept = None
        self.keep_ok = True       # False: behave like an engine whose logits did not fit

    def encode(self, lonlat):
        q = lonlat.numpy()
        e = O.encode(q, self.w, self.L)
        xq = np.zeros((q.shape[0], 4), np.float32)
        xq[:, :3] = O.query_xyz(q)
        return torch.from_numpy(e), torch.from_numpy(e.astype(np.float32)), torch.from_numpy(xq)

    def _logits(self, e32, xq):
        # (row by row: a query's logits must not depend on the batch it is computed in - BLAS picks
        # its blocking by shape - so that chunked and unchunked scans can be compared bit for bit)
        K, X = self.bank.keys.astype(np.float64), self.bank.xyz.astype(np.float64)
        e, x = e32.numpy().astype(np.float64), xq.numpy()[:, :3].astype(np.float64)
        s = np.stack([K @ r for r in e]) if len(e) else np.zeros((0, K.shape[0]))
        g = np.stack([X @ r for r in x]) if len(x) else np.zeros((0, X.shape[0]))
        return s, g

    def scan_stats(self, e32, xq, tau_sem, tau_geo, topk=0, keep_logits=False):
        self._kept = e32.clone() if keep_logits and self.keep_ok and not topk else None
        s, g = self._logits(e32, xq)
        # the engine's contract (range_hip.h): constant shift m = tau*log2(e), l = sum 2^(t - m)
        st = np.zeros((s.shape[0], 4), np.float64)
        t = s * tau_sem * LOG2E
        st[:, 0] = tau_sem * LOG2E; st[:, 1] = np.exp2(t - st[:, :1]).sum(1)
        if tau_geo > 0:
            t = g * tau_geo * LOG2E
            st[:, 2] = tau_geo * LOG2E; st[:, 3] = np.exp2(t - st[:, 2:3]).sum(1)
        else:
            st[:, 2] = -1e30
        st = torch.from_numpy(st.astype(np.float32))
        if not topk:
            return st
        tv, ti = O.topk64(s, topk)
        return st, torch.from_numpy(tv.astype(np.float32)), torch.from_numpy(ti + self.row_offset)

    # pass 1 in chunks of one scan (range_hip.h: range_scan_stats_at): the chunks' logits share one
    # workspace, addressed by attend_kept as if one call had kept them
    def p1_splits(self, n_queries):
        return 3

    def scan_stats_at(self, e32, xq, tau_sem, tau_geo, first_query, total_queries, n_splits=0):
        assert first_query % 64 == 0 and first_query + e32.shape[0] <= total_queries
        self.calls_at = getattr(self, "calls_at", 0) + 1
        if first_query == 0:
            self._kept = torch.zeros((total_queries, e32.shape[1]), dtype=e32.dtype) if self.keep_ok else None
            self._kept_n = 0
        if self._kept is not None and self._kept_n == first_query and self._kept.shape[0] == total_queries:
            self._kept[first_query:first_query + e32.shape[0]] = e32
            self._kept_n = first_query + e32.shape[0]
        else:
            self._kept = None
        kept, self._kept = self._kept, None
        st = self.scan_stats(e32, xq, tau_sem, tau_geo)        # (resets self._kept)
        self._kept = kept
        return st

    def merge_stats(self, parts):
        p = parts.numpy().astype(np.float64)
        out = np.zeros(p.shape[1:], np.float64)
        for c in (0, 2):
            m = p[:, :, c].max(0)
            out[:, c] = m
            out[:, c + 1] = (p[:, :, c + 1] * np.exp2(p[:, :, c] - m)).sum(0)
        return torch.from_numpy(out.astype(np.float32))

    def attend(self, e32, xq, tau_sem, tau_geo, beta, stats):
        s, g = self._logits(e32, xq)
        st = stats.numpy().astype(np.float64)
        w = beta * np.exp2(s * tau_sem * LOG2E - st[:, :1]) / st[:, 1:2]
        if tau_geo > 0:
            w = w + (1 - beta) * np.exp2(g * tau_geo * LOG2E - st[:, 2:3]) / st[:, 3:4]
        V = self.bank.values.astype(np.float64)
        return torch.from_numpy(np.stack([r @ V for r in w]).astype(np.float32))

    def kept_queries(self):
        if self._kept is None:
            return 0
        return getattr(self, "_kept_n", None) or self._kept.shape[0]

    def attend_kept(self, first, xq, tau_sem, tau_geo, beta, stats):
        assert first % 64 == 0 and first + xq.shape[0] <= self._kept.shape[0]
        return self.attend(self._kept[first:first + xq.shape[0]], xq, tau_sem, tau_geo, beta, stats)

    def blend(self, G, H, beta):
        return ((1.0 - beta) * G.double() + beta * H.double()).float()

    def finalize(self, partials, e64, out=None):
        acc = partials[0].clone()
        for p in partials[1:]:
            acc = acc + p
        res = torch.cat([acc.double(), e64], dim=1)
        if out is None:
            return res
        out.copy_(res)
        return out

    def topk_stream(self, e32, k):
        s, _ = self._logits(e32, torch.zeros((e32.shape[0], 4)))
        tv, ti = O.topk64(s, k)
        return torch.from_numpy(tv.astype(np.float32)), torch.from_numpy(ti + self.row_offset)

    def merge_topk(self, vals, idxs):
        W, B, k = vals.shape
        v = vals.permute(1, 0, 2).reshape(B, W * k).numpy()
        i = idxs.permute(1, 0, 2).reshape(B, W * k).numpy()
        order = np.lexsort((i, -v), axis=1)[:, :k]
        return (torch.from_numpy(np.take_along_axis(v, order, 1)),
                torch.from_numpy(np.take_along_axis(i, order, 1)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, B, L, H, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)          # (up to 8 rank processes on this box's 8 cores)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        locs, vals, keys = synth.make_bank(N, 11)
        full = O.prep_bank(locs, vals, keys)
        r0, r1 = shard_rows(N, world, rank)
        shard = O.Bank(full.keys[r0:r1], full.values[r0:r1], full.xyz[r0:r1])
        w = synth.make_encoder_weights(L, H, 256, 2, 5)
        q = synth.make_queries(B, seed=100 + rank)
        for name, beta, chunks, keep in (("RANGE+", 0.5, 1, True), ("RANGE+", 0.0, 3, True),
                                         ("RANGE", None, 2, True), ("RANGE+", 0.25, 3, False)):
            eng = OracleShardEngine(w, L, shard, r0)
            eng.keep_ok = keep           # False: the engine could not keep its logits
            model = ShardedRange(eng, name, beta, n_chunks=chunks)
            model.min_chunk = 2          # exercise the chunked, overlapped exchange on small batches
            assert len(model._chunk_bounds(B)) == min(chunks, max(1, B // 64))
            out = model(torch.from_numpy(q)).numpy()
            ref = O.forward(q, w, L, full, name, beta)      # unsharded oracle, own queries
            err = float(np.abs(out - ref).max())
            assert out.shape == (B, 1280) and err < 1e-5, (name, beta, err)   # f32 op-order noise of the reference
        # pass 1 per chunk, its gathers / all-reduces overlapped, == one pass 1 between blocking
        # collectives, bit for bit (same bank splits for every chunk; dist.ShardedRange._scan)
        outs = {}
        for chunked in (True, False):
            eng = OracleShardEngine(w, L, shard, r0)
            model = ShardedRange(eng, "RANGE+", 0.5, n_chunks=3)
            model.min_chunk = 2
            assert model.pass1_chunked
            model.pass1_chunked = chunked
            outs[chunked] = model(torch.from_numpy(q)).numpy()
            assert eng.calls_at == (3 if chunked else 1) and eng.kept_queries() == world * B
            sw = model.sweep(torch.from_numpy(q), (0.0, 1.0)).numpy()
            outs[chunked, "sweep"] = sw
            # bytes this rank moved: gathers of 1040 B per query to W-1 peers, 16 B of statistics per
            # scanned query to every peer, 4 KB of partials per own query from every peer (x3: forward + the sweep's two)
            model.reset_bytes()
            model(torch.from_numpy(q))
            assert model.bytes_sent["gather"] == B * 1040 * (world - 1)
            assert model.bytes_sent["exchange"] == B * 4096 * (world - 1) == model.bytes_received["exchange"]
            assert model.bytes_sent["reduce"] == world * B * 16 * (world - 1)
        assert np.array_equal(outs[True], outs[False]) and np.array_equal(outs[True, "sweep"], outs[False, "sweep"])
        # beta sweep: one scan, H and G once, every beta from them
        model = ShardedRange(OracleShardEngine(w, L, shard, r0), "RANGE+", 0.5, n_chunks=2)
        model.min_chunk = 2
        betas = (0.0, 0.25, 1.0)
        sw = model.sweep(torch.from_numpy(q), betas).numpy()
        for j, b in enumerate(betas):
            err = float(np.abs(sw[j] - O.forward(q, w, L, full, "RANGE+", b)).max())
            assert sw.shape == (3, B, 1280) and err < 1e-5, (b, err)
        model = ShardedRange(OracleShardEngine(w, L, shard, r0), "RANGE+", 0.5)
        tv, ti = model.topk(torch.from_numpy(q), 8)
        s, _ = O.logits64(O.encode(q, w, L), q, full)
        rv, ri = O.topk64(s, 8)
        assert np.array_equal(ti.numpy(), ri)
        np.testing.assert_allclose(tv.numpy(), rv, atol=1e-7)
        # forward(topk=k): the side channel of the SAME call (one encode, one gather of the operands, ONE
        # all-gather of candidates) == topk(), and the embeddings are those of the plain forward - chunked
        # (the own rows sit at W lo + rank (hi - lo) of every chunk) and not, blocking and not
        for chunks_n, blocking in ((1, False), (3, False), (2, True)):
            model = ShardedRange(OracleShardEngine(w, L, shard, r0), "RANGE+", 0.5, n_chunks=chunks_n)
            model.min_chunk = 2
            if blocking:
                model.blocking, model.pass1_chunked = True, False
            plain = model(torch.from_numpy(q)).numpy()
            model.reset_bytes()
            out, fv, fi = model.forward(torch.from_numpy(q), topk=8)
            assert np.array_equal(out.numpy(), plain)
            assert np.array_equal(fi.numpy(), ti.numpy()) and np.array_equal(fv.numpy(), tv.numpy())
            assert model.bytes_sent["topk"] == world * B * 8 * 12 * (world - 1)       # the one all-gather
            assert model.bytes_sent["gather"] == B * 1040 * (world - 1)              # the operands travelled once
        ret[rank] = "ok"
    except Exception as ex:  # noqa: BLE001
        import traceback
        ret[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_sharded_forward_gloo(world):
    """(world 8 = the north-star world size: 75-row shards, 7-peer exchanges, chunked == unchunked,
    byte counters - as eight gloo rank processes)"""
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), 601, 200, 10, 64, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


def test_shard_rows_cover():
    for n in (1, 7, 100_000):
        for w in (1, 2, 8):
            spans = [shard_rows(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def _ragged_worker(rank, world, port, ret):
    """embed / embed_sweep / embed_topk: every rank brings its OWN number of queries (one of them
    none at all), walked in outer steps shorter than the largest count."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, L, H = 601, 10, 64
        locs, vals, keys = synth.make_bank(N, 11)
        full = O.prep_bank(locs, vals, keys)
        r0, r1 = shard_rows(N, world, rank)
        shard = O.Bank(full.keys[r0:r1], full.values[r0:r1], full.xyz[r0:r1])
        w = synth.make_encoder_weights(L, H, 256, 2, 5)
        model = ShardedRange(OracleShardEngine(w, L, shard, r0), "RANGE+", 0.5)
        for counts, chunk in (((5, 0, 130), 64), ((1, 1, 0), None), ((0, 70, 0), 64), ((64, 64, 64), 32)):
            B = counts[rank]
            q = synth.make_queries(max(B, 1), seed=300 + rank)[:B]
            out = model.embed(torch.from_numpy(q), chunk=chunk).numpy()
            assert out.shape == (B, 1280)
            if B:
                err = float(np.abs(out - O.forward(q, w, L, full, "RANGE+", 0.5)).max())
                assert err < 1e-5, (counts, err)
            sw = model.embed_sweep(torch.from_numpy(q), (0.0, 1.0), chunk=chunk).numpy()
            assert sw.shape == (2, B, 1280)
            tv, ti = model.embed_topk(torch.from_numpy(q), 8, chunk=chunk)
            assert tuple(tv.shape) == (B, 8) and tuple(ti.shape) == (B, 8)
            out2, fv, fi = model.embed(torch.from_numpy(q), chunk=chunk, topk=8)
            assert np.array_equal(out2.numpy(), out) and np.array_equal(fi.numpy(), ti.numpy()) and np.array_equal(fv.numpy(), tv.numpy())
            if B:
                for j, b in enumerate((0.0, 1.0)):
                    assert float(np.abs(sw[j] - O.forward(q, w, L, full, "RANGE+", b)).max()) < 1e-5
                s, _ = O.logits64(O.encode(q, w, L), q, full)
                assert np.array_equal(ti.numpy(), O.topk64(s, 8)[1])
        # the communication buffers are one per name, not one per batch shape
        n_buf = len(model._bufs)
        for B in (3, 17, 40, 9):
            model.embed(torch.from_numpy(synth.make_queries(B, seed=B)))
        assert len(model._bufs) == n_buf
        ret[rank] = "ok"
    except Exception as ex:  # noqa: BLE001
        import traceback
        ret[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


def test_sharded_embed_ragged_batches_gloo():
    world = 3
    ret = mp.Manager().dict()
    mp.spawn(_ragged_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


def _layout_worker(rank, world, port, R, ret):
    """2-D layout R x Q (dist.make_layout): the bank row-sharded inside groups of R ranks, the groups
    independent; every rank's own queries against the WHOLE bank."""
    from range_amd.dist import make_layout
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, L, H, B = 601, 10, 64, 70 + rank
        locs, vals, keys = synth.make_bank(N, 11)
        full = O.prep_bank(locs, vals, keys)
        group, si, qg = make_layout(R)
        assert (si, qg) == (rank % R, rank // R) and (dist.get_world_size(group) == R)
        r0, r1 = shard_rows(N, R, si)
        shard = O.Bank(full.keys[r0:r1], full.values[r0:r1], full.xyz[r0:r1])
        w = synth.make_encoder_weights(L, H, 256, 2, 5)
        model = ShardedRange(OracleShardEngine(w, L, shard, r0), "RANGE+", 0.5, group=group)
        assert model.world == R and model.rank == si
        q = synth.make_queries(B, seed=400 + rank)
        out = model.embed(torch.from_numpy(q), chunk=64).numpy()
        err = float(np.abs(out - O.forward(q, w, L, full, "RANGE+", 0.5)).max())
        assert out.shape == (B, 1280) and err < 1e-5, err
        tv, ti = model.embed_topk(torch.from_numpy(q), 8)
        s, _ = O.logits64(O.encode(q, w, L), q, full)
        assert np.array_equal(ti.numpy(), O.topk64(s, 8)[1])
        ret[rank] = "ok"
    except Exception as ex:  # noqa: BLE001
        import traceback
        ret[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("R", [1, 2, 4])
def test_two_dimensional_layout_gloo(R):
    world = 4
    ret = mp.Manager().dict()
    mp.spawn(_layout_worker, args=(world, _free_port(), R, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


def _subgroup_layout_worker(rank, world, port, ret):
    """make_layout over a NON-default group: only the sub-group's ranks call it (rank 0 of the job
    stays out: it could not be made to enter the new_group calls)."""
    from range_amd.dist import make_layout
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sub = dist.new_group(ranks=list(range(1, world)))      # (every rank of the job enters this one)
        if rank >= 1:
            N, L, H, B, R = 601, 10, 64, 66, 2
            group, si, qg = make_layout(R, group=sub)
            assert (si, qg) == ((rank - 1) % R, (rank - 1) // R) and dist.get_world_size(group) == R
            locs, vals, keys = synth.make_bank(N, 11)
            full = O.prep_bank(locs, vals, keys)
            r0, r1 = shard_rows(N, R, si)
            w = synth.make_encoder_weights(L, H, 256, 2, 5)
            model = ShardedRange(OracleShardEngine(w, L, O.Bank(full.keys[r0:r1], full.values[r0:r1], full.xyz[r0:r1]), r0),
                                 "RANGE+", 0.5, group=group)
            q = synth.make_queries(B, seed=500 + rank)
            err = float(np.abs(model.embed(torch.from_numpy(q)).numpy() - O.forward(q, w, L, full, "RANGE+", 0.5)).max())
            assert err < 1e-5, err
        ret[rank] = "ok"
    except Exception as ex:  # noqa: BLE001
        import traceback
        ret[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


def test_layout_over_a_subgroup_of_the_job_gloo():
    world = 5
    ret = mp.Manager().dict()
    mp.spawn(_subgroup_layout_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


def test_embed_refuses_a_b_max_below_the_own_count():
    class _E:            # (never reached)
        pass
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        model = ShardedRange(_E(), "RANGE+", 0.5)
        with pytest.raises(ValueError, match="b_max"):
            model.embed(torch.zeros((5, 2), dtype=torch.float64), b_max=4)
    finally:
        dist.destroy_process_group()


# ---- world 8 beyond the plain forward: ragged and empty shares, every row_shards divisor (8x1, 4x2,
#      2x4, 1x8), the blocking escape hatch - eight rank THREADS of this process on torch's in-process
#      backend (tools/thread_ranks.py: the world size of the north star, at the price of one process)
from tools.thread_ranks import run_rank_threads, threaded_backend_available  # noqa: E402

needs_threads = pytest.mark.skipif(not threaded_backend_available(), reason="torch's threaded process group is not in this build")


def _w8_setup(rank, R=None, group=None):
    N, L, H = 601, 10, 64
    locs, vals, keys = synth.make_bank(N, 11)
    full = O.prep_bank(locs, vals, keys)
    world = dist.get_world_size(group)
    R = R or world
    si = dist.get_rank(group) % R
    r0, r1 = shard_rows(N, R, si)
    w = synth.make_encoder_weights(L, H, 256, 2, 5)
    return full, O.Bank(full.keys[r0:r1], full.values[r0:r1], full.xyz[r0:r1]), r0, w, L


def _w8_ragged(rank, world):
    full, shard, r0, w, L = _w8_setup(rank)
    model = ShardedRange(OracleShardEngine(w, L, shard, r0), "RANGE+", 0.5)
    model.min_chunk = 2
    # 1 250-query-like ragged shares scaled down: some ranks bring nothing, one brings far more
    for counts, chunk in (((5, 0, 130, 64, 0, 1, 77, 12), 64), ((0, 0, 0, 0, 0, 0, 0, 9), None), ((66,) * 8, 32)):
        B = counts[rank]
        q = synth.make_queries(max(B, 1), seed=300 + rank)[:B]
        out = model.embed(torch.from_numpy(q), chunk=chunk).numpy()
        assert out.shape == (B, 1280)
        if B:
            assert float(np.abs(out - O.forward(q, w, L, full, "RANGE+", 0.5)).max()) < 1e-5, counts
        tv, ti = model.embed_topk(torch.from_numpy(q), 8, chunk=chunk)
        assert tuple(ti.shape) == (B, 8)
        if B:
            s, _ = O.logits64(O.encode(q, w, L), q, full)
            assert np.array_equal(ti.numpy(), O.topk64(s, 8)[1])
    # equal shares: the byte counters of one forward at world 8 (7 peers)
    B = 128
    q = synth.make_queries(B, seed=40 + rank)
    model.reset_bytes()
    out = model(torch.from_numpy(q)).numpy()
    assert float(np.abs(out - O.forward(q, w, L, full, "RANGE+", 0.5)).max()) < 1e-5
    assert model.bytes_sent["gather"] == B * 1040 * 7 and model.bytes_sent["reduce"] == 8 * B * 16 * 7
    assert model.bytes_sent["exchange"] == B * 4096 * 7 == model.bytes_received["exchange"]
    # the blocking escape hatch gives the same bits
    blocking = ShardedRange(OracleShardEngine(w, L, shard, r0), "RANGE+", 0.5)
    blocking.min_chunk = 2
    blocking.blocking, blocking.pass1_chunked = True, False
    assert np.array_equal(blocking(torch.from_numpy(q)).numpy(), out)


def _w8_layout(rank, world, R):
    from range_amd.dist import make_layout
    group, si, qg = make_layout(R)
    assert (si, qg) == (rank % R, rank // R)
    full, shard, r0, w, L = _w8_setup(rank, R)
    model = ShardedRange(OracleShardEngine(w, L, shard, r0), "RANGE+", 0.5, group=group)
    assert model.world == R and model.rank == si
    B = 70 + rank
    q = synth.make_queries(B, seed=400 + rank)
    out = model.embed(torch.from_numpy(q), chunk=64).numpy()
    assert out.shape == (B, 1280) and float(np.abs(out - O.forward(q, w, L, full, "RANGE+", 0.5)).max()) < 1e-5
    sw = model.embed_sweep(torch.from_numpy(q), (0.0, 1.0)).numpy()
    for j, b in enumerate((0.0, 1.0)):
        assert float(np.abs(sw[j] - O.forward(q, w, L, full, "RANGE+", b)).max()) < 1e-5


@needs_threads
def test_world8_ragged_and_empty_shares_rank_threads():
    res = run_rank_threads(8, _w8_ragged)
    assert res == {r: "ok" for r in range(8)}, res


@needs_threads
@pytest.mark.parametrize("R", [1, 2, 4, 8])
def test_world8_every_layout_rank_threads(R):
    res = run_rank_threads(8, _w8_layout, R)
    assert res == {r: "ok" for r in range(8)}, res
