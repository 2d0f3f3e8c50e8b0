"""First contact with RCCL over xGMI, written so that it RUNS BY ITSELF the day a multi-GPU box is
there: world = min(visible GPUs, 8), skipped below 2, backend "nccl" (= RCCL), one fresh rank process
per GPU (the pytest process never touches a GPU: ``torch.cuda.device_count()`` does not initialise one
on this image).  What only this test exercises: RCCL's asynchronous collectives on its own stream
under the chunked schedule of range_amd/dist.py (gathers of all chunks behind the encoder, statistics of
chunk c and exchange of chunk c-1 travelling under pass 1 / pass 2 of their neighbours), sub-groups for
the 2-D layouts, device-resident buffers without host staging.

  C4 shape  RANGE+ beta=0.5, range_db_large row-sharded over ALL ranks, 100 000 queries in total (ragged:
            rank r brings one query more than rank r-1), every row through the planted-column properties,
            a sample per rank against the float64 oracle over the whole bank; chunked pass 1 == unchunked
            pass 1 bit for bit;
  C5 shape  the beta sweep over the sharded bank, a sample of every beta against the oracle;
  top-k     global top-16 with ONE all-gather, indices against the float64 oracle;
  layouts   every divisor R of the world as ``row_shards`` (R x W/R): the same queries, the same results
            within the split-order rounding of float32;
  drop-in   ``model(coords)`` with the same batch on every rank returns the full batch on every rank;
            the sharded ``save_embeddings`` moves every rank's rows to rank 0 only.

The same shapes run on every box as two gloo ranks sharing one GPU (tests/test_gpu_sharded.py).

``test_rccl_single_rank_stream_semantics`` needs ONE GPU: a one-rank RCCL group.  Its collectives move
nothing between devices, but they are ProcessGroupNCCL's - enqueued on the backend's own stream,
``work.wait()`` a stream dependency instead of a host wait - which neither gloo nor the rank threads
provide: the chunked, overlapped schedule of range_amd/dist.py (forced to 2 and 4 chunks) then runs
with real asynchrony between the compute stream and the collectives' stream, and a missing
dependency shows as a result that differs from the one-GPU model's or from its own repeat."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
L, H, SEED, N = 40, 512, 1234, 100_000


def _world():
    return min(torch.cuda.device_count(), 8)


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
    from oracle import range_oracle as O
    for j, b in enumerate(betas):
        got = out_rows[j]
        e = got[:, 1024:]
        low = np.abs(q_rows[:, 1]) <= 30
        np.testing.assert_allclose(e[low], O.encode(q_rows[low], w, L), rtol=0, atol=5e-9)
        np.testing.assert_allclose(got[:, :1024], O.retrieve64(e, q_rows, obank, "RANGE+", b), rtol=0, atol=2e-5)
        np.testing.assert_allclose(got, O.retrieve(e, q_rows, obank, "RANGE+", b), rtol=0, atol=1e-4)


def _rank(rank, world, port, ck, rbank, tmp, ret, backend="nccl", body=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (the product's timeout - 120 s, RANGE_DIST_TIMEOUT_S - not torch's 600: a rank that never arrives costs
    # two minutes of the first multi-GPU run of this suite, not ten)
    from datetime import timedelta
    from range_amd.dist import dist_timeout_s
    tmo = timedelta(seconds=dist_timeout_s())
    if backend == "nccl":
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
    else:       # rehearsal of THIS test's body on a one-GPU box: gloo ranks sharing cuda:0 (see the test below)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
    try:
        (body or _body)(rank, world, dev, ck, rbank, tmp, backend)
        ret[rank] = "ok"
    except Exception as ex:  # noqa: BLE001
        import traceback
        ret[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


def _body(rank, world, dev, ck, rbank, tmp, backend, total_queries=100_000):
    """What every rank does, whatever carries the collectives (RCCL; gloo ranks sharing a GPU; the rank
    THREADS of tests/test_gpu_world8.py).  ``total_queries``: the C4 batch over all ranks."""
    import torch.distributed as dist
    from oracle import range_oracle as O
    from range_amd import load_model
    from range_amd.save import save_embeddings
    from tools import synth
    if True:
        assert dist.get_backend() == backend
        obank = O.prep_bank(*_bank_arrays())
        w = synth.make_encoder_weights(L, H, 256, 2, SEED)
        vmin, vmax = float(obank.values.min()), float(obank.values.max())
        m = load_model("RANGE+", pretrained_path=ck, device=dev, db_path=rbank, beta=0.5, shards=world)
        assert m.row_range == ((N * rank) // world, (N * (rank + 1)) // world)
        # ---- C4 shape: 100 000 queries over the ranks, ragged, pole to pole
        B = total_queries // world + rank
        q = synth.make_queries(B, seed=70 + rank, lat_max=90.0)
        x = torch.from_numpy(q).to(dev)
        m.sharded.comm_timing(True)
        out = m(x, local=True, return_device=True)
        comm = m.sharded.comm_timing(False)
        assert set(comm) == {"gather", "reduce", "exchange", "total"} and comm["total"] >= 0.0
        assert out.shape == (B, 1280) and bool(torch.isfinite(out).all())
        assert m.engine.kept_queries() > 0               # pass 2 ran on the kept logits of the shard
        assert float((out[:, 0] - 1.0).abs().max()) < 1e-5 and float((out[:, 1] + 2.5).abs().max()) < 2.5e-5
        assert float(out[:, :1024].max()) <= vmax and float(out[:, :1024].min()) >= vmin
        assert float((out[:, 1024:].norm(dim=1) - 1.0).abs().max()) < 1e-12
        idx = np.sort(np.random.default_rng(rank).choice(B, 128, replace=False))
        sel = torch.from_numpy(idx).to(dev)
        _sample_check(out[sel].cpu().numpy()[None], q[idx], obank, w, (0.5,))
        # chunked pass 1 (collectives under compute) == one pass 1 between blocking collectives, bit for bit
        Bs = 10_000 // world // 64 * 64 or 64
        chunked = m.sharded.forward(x[:Bs].contiguous())
        m.sharded.pass1_chunked = False
        try:
            plain = m.sharded.forward(x[:Bs].contiguous())
        finally:
            m.sharded.pass1_chunked = True
        assert torch.equal(chunked, plain)
        # ---- top-k: one all-gather of the shards' candidates
        tv, ti = m.topk(x[sel], 16, local=True)
        e = out[sel, 1024:].cpu().numpy()
        rv, ri = O.topk64(O.logits64(e, q[idx], obank)[0], 16)
        assert int((ti.cpu().numpy() != ri).any(axis=1).sum()) <= 1        # (ties within 4 ulp of float32)
        np.testing.assert_allclose(tv.cpu().numpy(), rv, rtol=0, atol=3e-7)
        # ---- C5 shape: the beta sweep
        betas = (0.0, 0.25, 0.5, 0.75, 1.0)
        sw = m.sweep(x[:Bs], betas, local=True, return_device=True)
        assert sw.shape == (5, Bs, 1280) and bool(torch.isfinite(sw).all())
        si = np.sort(np.random.default_rng(10 + rank).choice(Bs, 64, replace=False))
        _sample_check(sw[:, torch.from_numpy(si).to(dev)].cpu().numpy(), q[si], obank, w, betas)
        ref_rows = out[:Bs].clone()
        del out, sw
        # ---- every 2-D layout R x W/R: the same rows within float32 split-order rounding
        for R in [r for r in range(1, world) if world % r == 0]:
            m2 = load_model("RANGE+", pretrained_path=ck, device=dev, db_path=rbank, beta=0.5, shards=world,
                            row_shards=R)
            assert m2.row_shards == R and m2.engine.n_rows in (N // R, N // R + 1)
            o2 = m2(x[:Bs], local=True, return_device=True)
            d = (o2 - ref_rows).abs()
            assert float(d[:, 2:].max()) < 2e-6 and float(d[:, :2].max()) < 5e-5, (R, float(d.max()))
            del m2, o2
        # ---- drop-in: the same batch on every rank, the full result on every rank
        qf = synth.make_queries(1001, seed=5)
        full = m(torch.from_numpy(qf))
        assert isinstance(full, np.ndarray) and full.shape == (1001, 1280) and full.dtype == np.float64
        _sample_check(full[None, :64], qf[:64], obank, w, (0.5,))
        _sample_check(full[None, -64:], qf[-64:], obank, w, (0.5,))
        # ... with the top-k side channel from the same call (round 6): one encode, one gather of the operands,
        # ONE all-gather of the shards' candidates - model.topk()'s result bit for bit
        full2, fv, fi = m(torch.from_numpy(qf), return_topk=16)
        tv2, ti2 = m.topk(torch.from_numpy(qf), 16)
        assert np.array_equal(full2, full) and torch.equal(fi, ti2) and torch.equal(fv, tv2)
        # ---- save_embeddings over the sharded model: every rank embeds its rows, rank 0 receives and writes
        from argparse import Namespace

        def loader(n_batches, bs):
            for i in range(n_batches):
                n = bs if i + 1 < n_batches else bs - 37
                yield torch.from_numpy(synth.make_queries(n, seed=900 + i)), torch.arange(n, dtype=torch.float32)
        a = Namespace(embeddings_dir=os.path.join(tmp, "emb"), location_model_name="RANGE+", task_name="t")
        save_embeddings(a, loader(3, 500), loader(2, 300), m)
        dist.barrier()
        z = np.load(os.path.join(tmp, "emb", "RANGE+", "t_train.npz"))
        assert z["embeddings"].shape == (1463, 1280)
        _sample_check(z["embeddings"][None, :32], z["coords"][:32], obank, w, (0.5,))


def test_rccl_row_sharded_at_c4_c5_shapes(tmp_path):
    world = _world()
    backend = "nccl"
    # RANGE_RCCL_TEST_REHEARSAL=W: the body of this test with W gloo ranks sharing cuda:0 - how the test
    # itself was checked on the one-GPU boxes it was written on (every assertion below has run green
    # that way with W = 2 and W = 4; what the rehearsal cannot reach is RCCL itself)
    if os.environ.get("RANGE_RCCL_TEST_REHEARSAL"):
        world, backend = int(os.environ["RANGE_RCCL_TEST_REHEARSAL"]), "gloo"
    if world < 2:
        pytest.skip(f"RCCL needs >= 2 GPUs, {torch.cuda.device_count()} visible "
                    "(the same shapes run as two gloo ranks on one GPU: tests/test_gpu_sharded.py)")
    from range_amd.bank import prepare_bank
    from range_amd.bankfile import write_bankfile
    from tools import synth
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=SEED)
    rbank = write_bankfile(str(tmp_path / "large.rbank"), prepare_bank(*_bank_arrays()))
    ret = mp.Manager().dict()
    mp.spawn(_rank, args=(world, _free_port(), ck, rbank, str(tmp_path), ret, backend), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


def _body_single(rank, world, dev, ck, rbank, tmp, backend):
    """One RCCL rank: the sharded model over the whole bank against the plain one-GPU model."""
    import torch.distributed as dist
    from oracle import range_oracle as O
    from range_amd import load_model
    from range_amd.save import save_embeddings
    from tools import synth
    assert world == 1 and dist.get_backend() == "nccl"
    obank = O.prep_bank(*_bank_arrays())
    w = synth.make_encoder_weights(L, H, 256, 2, SEED)
    plain = load_model("RANGE+", pretrained_path=ck, device=dev, db_path=rbank, beta=0.5)
    m = load_model("RANGE+", pretrained_path=ck, device=dev, db_path=rbank, beta=0.5, shards=1)
    B = 20_000 + 37
    q = synth.make_queries(B, seed=71, lat_max=90.0)
    x = torch.from_numpy(q).to(dev)
    ref = plain(x, return_device=True)
    outs = {}
    for n_chunks in (1, 2, 4):
        m.sharded.n_chunks = n_chunks
        m.sharded.comm_timing(True)
        out = m(x, local=True, return_device=True)
        comm = m.sharded.comm_timing(False)
        assert comm["total"] >= 0.0
        for _ in range(2):            # a race between the two streams would not repeat itself bit for bit
            assert torch.equal(m(x, local=True, return_device=True), out), n_chunks
        d = (out - ref).abs()
        assert float(d[:, 1024:].max()) == 0.0                     # the encoder does not know about shards
        assert float(d[:, 2:1024].max()) < 2e-6 and float(d[:, :2].max()) < 5e-5, (n_chunks, float(d.max()))
        outs[n_chunks] = out
    # the chunking changes nothing: same bank splits, statistics merged in rank order
    Bs = 9984
    m.sharded.n_chunks = 4
    chunked = m.sharded.forward(x[:Bs].contiguous())
    m.sharded.pass1_chunked = False
    try:
        unchunked = m.sharded.forward(x[:Bs].contiguous())
    finally:
        m.sharded.pass1_chunked = True
    assert torch.equal(chunked, unchunked)
    idx = np.sort(np.random.default_rng(3).choice(B, 96, replace=False))
    sel = torch.from_numpy(idx).to(dev)
    _sample_check(outs[4][sel].cpu().numpy()[None], q[idx], obank, w, (0.5,))
    # top-k and the beta sweep through the same collectives
    tv, ti = m.topk(x[sel], 16, local=True)
    pv, pi = plain.topk(x[sel], 16)
    assert torch.equal(ti.cpu(), torch.as_tensor(pi).cpu()) and torch.equal(tv.cpu(), torch.as_tensor(pv).cpu())
    betas = (0.0, 0.5, 1.0)
    m.sharded.n_chunks = 2
    sw = m.sweep(x[:Bs], betas, local=True, return_device=True)
    si = np.sort(np.random.default_rng(4).choice(Bs, 48, replace=False))
    _sample_check(sw[:, torch.from_numpy(si).to(dev)].cpu().numpy(), q[si], obank, w, betas)
    # the batch driver: results travel to rank 0 (itself) behind the next batch's compute
    from argparse import Namespace

    def loader(n_batches, bs):
        for i in range(n_batches):
            n = bs if i + 1 < n_batches else bs - 37
            yield torch.from_numpy(synth.make_queries(n, seed=900 + i)), torch.arange(n, dtype=torch.float32)
    a = Namespace(embeddings_dir=os.path.join(tmp, "emb1"), location_model_name="RANGE+", task_name="t")
    save_embeddings(a, loader(4, 1500), loader(2, 300), m)
    z = np.load(os.path.join(tmp, "emb1", "RANGE+", "t_train.npz"))
    assert z["embeddings"].shape == (4 * 1500 - 37, 1280)
    want = plain(torch.from_numpy(z["coords"]).to(dev), return_device=True).cpu().numpy()
    np.testing.assert_allclose(z["embeddings"], want, rtol=0, atol=5e-5)
    # (the encoder's kernels differ by batch size - 1 500 per batch there, 5 963 at once here: float64 rounding)
    np.testing.assert_allclose(z["embeddings"][:, 1024:], want[:, 1024:], rtol=0, atol=1e-13)


def test_rccl_single_rank_stream_semantics(tmp_path):
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    from range_amd.bank import prepare_bank
    from range_amd.bankfile import write_bankfile
    from tools import synth
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=SEED)
    rbank = write_bankfile(str(tmp_path / "large.rbank"), prepare_bank(*_bank_arrays()))
    ret = mp.Manager().dict()
    mp.spawn(_rank, args=(1, _free_port(), ck, rbank, str(tmp_path), ret, "nccl", _body_single), nprocs=1, join=True)
    assert dict(ret) == {0: "ok"}, dict(ret)
