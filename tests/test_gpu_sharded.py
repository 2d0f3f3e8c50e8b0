"""The row-sharded PRODUCT entry at BASELINE's multi-GPU shapes, on what a one-GPU box offers: two
rank processes with the real engine sharing cuda:0, each holding half of range_db_large, collectives
over gloo (staged through the host, range_amd/dist.py) - ``load_model(..., shards=2)``:

  C4 shape  RANGE+ beta=0.5, range_db_large row-sharded, 100 000 queries (50 000 per rank, ragged
            by one, in outer steps of 8 192 per rank): every row through the planted-column
            properties, a 256-query sample per rank against the float64 oracle over the WHOLE bank;
  C5 shape  the beta sweep {0, .25, .5, .75, 1} over the full bank, 20 000 queries: every beta's sample
            against the oracle, the beta = 0.5 slice against the plain forward;
  drop-in   ``model(coords)`` with the same batch on every rank returns the full batch (B = 1 001, 1);
            ``save_embeddings`` driven by the sharded model: every rank embeds its rows, only rank 0 receives
            results (byte counters) and writes the reference's files - the single-GPU driver's files;
  pass 1    per chunk with overlapped collectives == unchunked, bit for bit.

What only a real multi-GPU node can add is RCCL itself and the timing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
L, H, SEED, N = 40, 512, 1234, 100_000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bank_arrays():
    from tools import synth
    locs, vals, keys = synth.make_bank(N, 2024)
    vals = vals.copy()
    vals[:, 0] = 1.0            # constant columns: reproduced iff the weights of a row sum to one over ALL shards
    vals[:, 1] = -2.5
    return locs, vals, keys


def _sample_check(out_rows, q_rows, obank, w, betas):
    """out_rows: (nb, n, 1280) device results of the sampled queries for each beta."""
    from oracle import range_oracle as O
    for j, b in enumerate(betas):
        got = out_rows[j]
        e = got[:, 1024:]
        low = np.abs(q_rows[:, 1]) <= 30
        np.testing.assert_allclose(e[low], O.encode(q_rows[low], w, L), rtol=0, atol=5e-9)
        np.testing.assert_allclose(got[:, :1024], O.retrieve64(e, q_rows, obank, "RANGE+", b), rtol=0, atol=2e-5)
        np.testing.assert_allclose(got, O.retrieve(e, q_rows, obank, "RANGE+", b), rtol=0, atol=1e-4)


def _rank(rank, world, port, ck, rbank, tmp, ret):
    import torch.distributed as dist
    from oracle import range_oracle as O
    from range_amd import load_model
    from tools import synth
    from range_amd.save import save_embeddings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=rbank, beta=0.5, shards=world)
        assert m.row_range == ((N * rank) // world, (N * (rank + 1)) // world) and m.engine.n_rows == N // world
        obank = O.prep_bank(*_bank_arrays())
        w = synth.make_encoder_weights(L, H, 256, 2, SEED)
        vmin, vmax = float(obank.values.min()), float(obank.values.max())
        # ---- C4 shape: 100 000 queries over the two ranks (one more on rank 1: ragged), pole to pole
        B = 50_000 + rank
        q = synth.make_queries(B, seed=70 + rank, lat_max=90.0)
        x = torch.from_numpy(q).to("cuda:0")
        m.sharded.scan_chunk = 16384                      # 8 192 queries per rank and step
        out = m(x, local=True, return_device=True)
        assert out.shape == (B, 1280) and bool(torch.isfinite(out).all())
        assert m.engine.kept_queries() > 0               # pass 2 ran on the kept logits of the shard
        assert float((out[:, 0] - 1.0).abs().max()) < 1e-5 and float((out[:, 1] + 2.5).abs().max()) < 2.5e-5
        assert float(out[:, :1024].max()) <= vmax and float(out[:, :1024].min()) >= vmin
        assert float((out[:, 1024:].norm(dim=1) - 1.0).abs().max()) < 1e-12
        idx = np.sort(np.random.default_rng(rank).choice(B, 256, replace=False))
        _sample_check(out[torch.from_numpy(idx).cuda()].cpu().numpy()[None], q[idx], obank, w, (0.5,))
        # global top-16 of the sample: one all-gather of the shards' candidates
        tv, ti = m.topk(x[torch.from_numpy(idx).cuda()], 16, local=True)
        e = out[torch.from_numpy(idx).cuda(), 1024:].cpu().numpy()
        rv, ri = O.topk64(O.logits64(e, q[idx], obank)[0], 16)
        bad = np.nonzero((ti.cpu().numpy() != ri).any(axis=1))[0]
        assert len(bad) <= 1
        np.testing.assert_allclose(tv.cpu().numpy(), rv, rtol=0, atol=3e-7)
        del out
        # ---- C5 shape: the beta sweep over the full (sharded) bank
        betas = (0.0, 0.25, 0.5, 0.75, 1.0)
        Bs = 10_000
        sw = m.sweep(x[:Bs], betas, local=True, return_device=True)
        assert sw.shape == (5, Bs, 1280) and bool(torch.isfinite(sw).all())
        plain = m(x[:Bs], local=True, return_device=True)
        # beta = 0.5: the blend of H and G (range.py:238, float32) against the forward's one combined
        # weight (planted constant columns: float32 sums of 10^5 same-sign terms, formed once per
        # retrieval here and once for the combined weight there; other columns 1e-7)
        d = (sw[2] - plain).abs()
        assert float(d[:, 2:].max()) < 2e-6 and float(d[:, :2].max()) < 5e-5
        idx = np.sort(np.random.default_rng(10 + rank).choice(Bs, 128, replace=False))
        _sample_check(sw[:, torch.from_numpy(idx).cuda()].cpu().numpy(), q[idx], obank, w, betas)
        del sw, plain
        # ---- drop-in semantics: the same batch on every rank, the full result on every rank
        for Bf in (1001, 1):
            qf = synth.make_queries(Bf, seed=5)
            full = m(torch.from_numpy(qf))
            assert isinstance(full, np.ndarray) and full.shape == (Bf, 1280) and full.dtype == np.float64
            k = min(Bf, 64)
            _sample_check(full[None, :k], qf[:k], obank, w, (0.5,))
            if Bf > 1:
                _sample_check(full[None, -k:], qf[-k:], obank, w, (0.5,))
        # ---- save_embeddings over the sharded model (reference: range/utils/save.py:7-58): rank 0 writes
        from argparse import Namespace
        rng = np.random.default_rng(3)
        def loader(n_batches, bs):
            for i in range(n_batches):
                n = bs if i + 1 < n_batches else bs - 37                 # ragged tail (save.py:24-37)
                yield torch.from_numpy(synth.make_queries(n, seed=900 + i)), torch.arange(n, dtype=torch.float32)
        a = Namespace(embeddings_dir=os.path.join(tmp, "emb"), location_model_name="RANGE+", task_name="t")
        m.sharded.reset_bytes()
        save_embeddings(a, loader(3, 500), loader(2, 300), m)
        dist.barrier()
        # every rank embedded ITS rows of every batch and only rank 0 received results: 10 KB per own
        # query sent by the others (batches of 500, 500, 463 and 300, 263 rows over the ranks), nothing
        # by rank 0 - an all-gather of the full batch on every rank would move W times that
        per = sum((n + world - 1) // world for n in (500, 500, 463, 300, 263))
        assert m.sharded.bytes_sent["results"] == (0 if rank == 0 else per * 10240)
        assert m.sharded.bytes_received["results"] == ((world - 1) * per * 10240 if rank == 0 else 0)
        z = np.load(os.path.join(tmp, "emb", "RANGE+", "t_train.npz"))
        assert z["embeddings"].shape == (1463, 1280) and z["coords"].shape == (1463, 2) and z["y"].shape == (1463,)
        _sample_check(z["embeddings"][None, :32], z["coords"][:32], obank, w, (0.5,))
        zv = np.load(os.path.join(tmp, "emb", "RANGE+", "t_val.npz"))
        assert zv["embeddings"].shape == (563, 1280)
        # a 10 000-query batch: <= 10 000 / W rows (+ 1) of 10 KB leave a rank
        m.sharded.reset_bytes()
        big = Namespace(embeddings_dir=os.path.join(tmp, "emb_big"), location_model_name="RANGE+", task_name="b")
        save_embeddings(big, loader(1, 10_037), loader(1, 101), m)
        assert m.sharded.bytes_sent["results"] <= (10_000 // world + 1 + 64 // world + 1) * 10240
        dist.barrier()
        if rank == 0:
            # the same files from the single-GPU driver (whole bank on one GPU): coords and y equal, the
            # e-hat half to float64 rounding, the retrieval half within the float32 split-order rounding
            m1 = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=rbank, beta=0.5)
            a1 = Namespace(embeddings_dir=os.path.join(tmp, "emb1"), location_model_name="RANGE+", task_name="t")
            save_embeddings(a1, loader(3, 500), loader(2, 300), m1)
            for name, zs in (("t_train.npz", z), ("t_val.npz", zv)):
                z1 = np.load(os.path.join(tmp, "emb1", "RANGE+", name))
                assert np.array_equal(z1["coords"], zs["coords"]) and np.array_equal(z1["y"], zs["y"])
                # (e-hat: the encoder splits its first layer's K by batch size - last-bit differences)
                assert float(np.abs(z1["embeddings"][:, 1024:] - zs["embeddings"][:, 1024:]).max()) < 1e-13
                d = np.abs(z1["embeddings"][:, :1024] - zs["embeddings"][:, :1024])
                assert float(d[:, 2:].max()) < 2e-6 and float(d[:, :2].max()) < 5e-5
            del m1
        dist.barrier()
        # ---- pass 1 per chunk with its collectives under compute == one pass 1 between blocking
        #      collectives, bit for bit (the same bank splits for every chunk, statistics merged in rank order)
        xs = x[:4096].contiguous()
        m.sharded.n_chunks, m.sharded.min_chunk = 4, 256
        chunked = m.sharded.forward(xs)
        swc = m.sharded.sweep(xs, (0.0, 1.0))
        assert len(m.sharded._chunk_bounds(4096)) == 4 and m.engine.kept_queries() == world * 4096
        m.sharded.pass1_chunked = False
        plain = m.sharded.forward(xs)
        swp = m.sharded.sweep(xs, (0.0, 1.0))
        m.sharded.pass1_chunked = True
        assert torch.equal(chunked, plain) and torch.equal(swc, swp)
        ii = np.sort(np.random.default_rng(20 + rank).choice(4096, 64, replace=False))
        _sample_check(chunked[torch.from_numpy(ii).cuda()].cpu().numpy()[None], q[ii], obank, w, (0.5,))
        ret[rank] = "ok"
    except Exception as ex:  # noqa: BLE001
        import traceback
        ret[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


def test_sharded_product_entry_at_c4_c5_shapes(tmp_path):
    from tools import synth
    from range_amd.bank import prepare_bank
    from range_amd.bankfile import write_bankfile
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=SEED)
    rbank = write_bankfile(str(tmp_path / "large.rbank"), prepare_bank(*_bank_arrays()))
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_rank, args=(world, _free_port(), ck, rbank, str(tmp_path), ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


def test_load_model_shards_needs_a_process_group(tmp_path):
    from range_amd import load_model
    from tools import synth
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=10, hidden=64, seed=1)
    db = synth.write_bank(str(tmp_path / "db.npz"), 300, seed=3)
    with pytest.raises(RuntimeError, match="torch.distributed"):
        load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db, shards=2)
    with pytest.raises(ValueError, match="shards="):
        load_model("SatCLIP", pretrained_path=ck, device="cuda:0", shards=2)
