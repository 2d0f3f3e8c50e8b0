"""Two ranks, the REAL engine, one GPU: every rank is a process with its own context on cuda:0
holding half of the bank rows; torch.distributed runs over gloo (collectives staged through the
host, range_amd/dist.py) because RCCL needs one GPU per rank and the test box has one.  This is
the row-sharded path with world_size > 1 on the hand-written kernels: shard row offsets, merged
statistics, kept-logit offsets of the chunks, partial sums across shards, global top-k."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, ret):
    import torch.distributed as dist
    from oracle import range_oracle as O
    from range_amd import _native
    from tools import synth
    from range_amd.bank import prepare_bank
    from range_amd.dist import ShardedRange, shard_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, B, L, H = 3001, 200, 10, 64
        locs, vals, keys = synth.make_bank(N, 11)
        full = O.prep_bank(locs, vals, keys)
        bank = prepare_bank(locs, vals, keys)
        r0, r1 = shard_rows(N, world, rank)
        sh = bank.rows(r0, r1)
        w = synth.make_encoder_weights(L, H, 256, 2, 5)
        eng = _native.HipEngine("cuda:0")
        eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC,
                        [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                        [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])
        eng.set_bank(sh.keys, sh.values, sh.xyz, r0)
        q = synth.make_queries(B, seed=100 + rank)
        x = torch.from_numpy(q).to("cuda:0")
        for name, beta, chunks in (("RANGE+", 0.5, 1), ("RANGE+", 0.25, 3), ("RANGE", None, 2)):
            model = ShardedRange(eng, name, beta, n_chunks=chunks)
            model.min_chunk = 2
            out = model(x).cpu().numpy()
            assert eng.kept_queries() == world * B          # pass 2 ran on the kept logits
            ref = O.forward(q, w, L, full, name, beta)      # unsharded oracle, own queries
            err = float(np.abs(out - ref).max())
            assert out.shape == (B, 1280) and err < 2e-5, (name, beta, chunks, err)
        model = ShardedRange(eng, "RANGE+", 0.5, n_chunks=2)
        model.min_chunk = 2
        sw = model.sweep(x, (0.0, 1.0)).cpu().numpy()
        for j, b in enumerate((0.0, 1.0)):
            assert float(np.abs(sw[j] - O.forward(q, w, L, full, "RANGE+", b)).max()) < 2e-5
        tv, ti = ShardedRange(eng, "RANGE+", 0.5).topk(x, 8)
        s, _ = O.logits64(O.encode(q, w, L), q, full)
        rv, ri = O.topk64(s, 8)
        assert np.array_equal(ti.cpu().numpy(), ri)
        np.testing.assert_allclose(tv.cpu().numpy(), rv, atol=3e-7)
        ret[rank] = "ok"
    except Exception as ex:  # noqa: BLE001
        import traceback
        ret[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


def test_two_ranks_real_engine_one_gpu():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_rank, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)
