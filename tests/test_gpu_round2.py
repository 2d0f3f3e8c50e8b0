"""GPU tests of the round-2 paths over ragged shapes: the host-array contract
(range_forward_host), the small-batch two-kernel encoder, the reference-faithful spherical
harmonics at several degrees, the persistent stream top-k on small / ragged banks."""
import os

import numpy as np
import pytest
import torch

from oracle import range_oracle as O
from range_amd import _native, sh_table, synth
from range_amd.bank import prepare_bank

pytestmark = pytest.mark.gpu


def _weights(L, H, layers, seed):
    w = synth.make_encoder_weights(L, H, 256, layers, seed)
    ws = [w[f"layers.{i}.weight"] for i in range(layers)] + [w["last_layer.weight"]]
    bs = [w[f"layers.{i}.bias"] for i in range(layers)] + [w["last_layer.bias"]]
    return w, ws, bs


@pytest.mark.parametrize("B", [1, 63, 64, 1000, 4095, 4096, 5001, 9000])
def test_forward_host_equals_device_result(B):
    """The host contract: same values as the device-resident result (bit-identical below the
    two-part threshold, split-order rounding above), caller's array rows filled exactly once."""
    N, L, H = 5000, 10, 64
    w, ws, bs = _weights(L, H, 2, 5)
    bank = prepare_bank(*synth.make_bank(N, 3))
    eng = _native.HipEngine("cuda:0")
    eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, ws, bs, sh_table=sh_table.generate_table(L))
    eng.set_bank(bank.keys, bank.values, bank.xyz)
    x = torch.from_numpy(synth.make_queries(B, seed=B, lat_max=90.0)).cuda()
    dev = eng.forward(x, _native.MODEL_RANGE_PLUS, 0.3).cpu().numpy()
    out = np.full((B + 2, 1280), -7.0)
    got = eng.forward_host(x, _native.MODEL_RANGE_PLUS, 0.3, out=out[1:-1])
    assert got.base is out or got is out[1:-1] or np.shares_memory(got, out)
    assert (out[0] == -7.0).all() and (out[-1] == -7.0).all()          # nothing beyond the rows given
    if B < 4096:
        assert np.array_equal(out[1:-1], dev)
    else:
        np.testing.assert_allclose(out[1:-1], dev, rtol=1e-5, atol=2e-6)
        assert np.array_equal(out[1:-1, 1024:], dev[:, 1024:])
    fresh = eng.forward_host(x, _native.MODEL_RANGE, 1.0)               # a pool array, RANGE model
    assert fresh.shape == (B, 1280) and fresh.dtype == np.float64 and fresh.flags.c_contiguous
    np.testing.assert_allclose(fresh, eng.forward(x, _native.MODEL_RANGE, 1.0).cpu().numpy(), rtol=1e-5, atol=2e-6)
    with pytest.raises(ValueError):
        eng.forward_host(x, _native.MODEL_RANGE, 1.0, out=np.zeros((B, 1280), np.float32))


@pytest.mark.parametrize("L,H,layers,mode", [(40, 512, 2, "analytic"), (40, 256, 2, "analytic"), (16, 128, 3, "analytic"),
                                             (10, 64, 2, "closed-form"), (33, 320, 2, "analytic"), (7, 192, 1, "analytic"),
                                             (40, 384, 2, "analytic")])
def test_small_batch_encoder_split_matches_one_kernel(L, H, layers, mode, monkeypatch):
    """Batches of up to 2048 queries run the encoder as (first layer per column part and K range) +
    [second layer per column part, from 32 tiles on] + (rest):
    same embeddings as the one-kernel encoder to float64 summation-order noise, for every width
    the split exists for (and unchanged behaviour for those it does not), in both SH evaluations."""
    w, ws, bs = _weights(L, H, layers, 11)
    sh = _native.SH_ANALYTIC if mode == "analytic" else _native.SH_CLOSED_FORM
    for table in ([None, sh_table.generate_table(L)] if mode == "analytic" else [None]):
        monkeypatch.setenv("RANGE_ENC_SPLIT", "0")
        one = _native.HipEngine("cuda:0")
        monkeypatch.delenv("RANGE_ENC_SPLIT")
        two = _native.HipEngine("cuda:0")
        for e in (one, two):
            e.set_encoder(L, H, layers, 256, sh, ws, bs, sh_table=table)
        for B in (1, 15, 16, 17, 250, 600, 1024, 2048, 2049, 3000):
            x = torch.from_numpy(synth.make_queries(B, seed=B + L, lat_max=90.0)).cuda()
            a64, a32, axq = one.encode(x)
            b64, b32, bxq = two.encode(x)
            assert float((a64 - b64).abs().max()) < 5e-13
            assert torch.equal(axq, bxq)
            raw1, raw2 = one.encode_raw(x), two.encode_raw(x)
            assert float((raw1 - raw2).abs().max()) < 5e-12 * max(1.0, float(raw1.abs().max()))
        if table is None:
            x = torch.from_numpy(synth.make_queries(300, seed=1, lat_max=89.0)).cuda()
            np.testing.assert_allclose(two.encode(x)[0].cpu().numpy(), O.encode(x.cpu().numpy(), w, L, mode),
                                       rtol=0, atol=2e-12)


@pytest.mark.parametrize("L", [1, 2, 5, 10, 16, 25, 33, 40])
def test_reference_sh_tables_of_other_degrees(L):
    """The generated-polynomial evaluation for any degree L <= 40 (the table of L is a prefix of
    the table of 40): the kernel agrees with the CPU evaluation of the same table, and - the
    polynomials being well-conditioned for small L - with the exact basis too."""
    H = 64
    w, ws, bs = _weights(L, H, 2, 3)
    table = sh_table.generate_table(L)
    eng = _native.HipEngine("cuda:0")
    eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, ws, bs, sh_table=table)
    q = synth.make_queries(700, seed=L, lat_max=90.0)
    e = eng.encode(torch.from_numpy(q).cuda())[0].cpu().numpy()
    ref = O.encode(q, w, L, features=table.evaluate(q))
    band = np.abs(q[:, 1]) <= 45
    assert np.abs(e - ref)[band].max() < 1e-7 and np.abs(e - ref).max() < (1e-9 if L <= 16 else 2e-3)
    if L <= 16:
        np.testing.assert_allclose(e, O.encode(q, w, L), rtol=0, atol=1e-9)


@pytest.mark.parametrize("N,B,k", [(1, 1, 1), (15, 16, 4), (16, 17, 16), (17, 31, 16), (4095, 32, 16), (4097, 33, 7),
                                   (16385, 64, 16), (100_003, 20, 16), (70_000, 160, 16), (1029, 96, 3)])
def test_stream_topk_ragged_banks(N, B, k):
    """Persistent stream kernel on banks smaller than the grid, ragged last tiles, more than one
    supergroup of query groups: indices and values equal to the float64 oracle's."""
    rng = np.random.default_rng(N + B)
    keys = rng.standard_normal((N, 256)).astype(np.float32)
    keys /= np.linalg.norm(keys, axis=1, keepdims=True)
    qs = rng.standard_normal((B, 256)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    eng = _native.HipEngine("cuda:0")
    eng.set_bank(keys, np.zeros((N, 1024), np.float32), np.zeros((N, 3), np.float32), 1000)   # row offset
    tv, ti = eng.topk_stream(torch.from_numpy(qs).cuda(), k)
    s64 = qs.astype(np.float64) @ keys.astype(np.float64).T
    kk = min(k, N)
    rv, ri = O.topk64(s64, kk)
    tv, ti = tv.cpu().numpy(), ti.cpu().numpy()
    np.testing.assert_allclose(tv[:, :kk], rv, rtol=0, atol=4e-7)
    bad = np.nonzero((ti[:, :kk] - 1000 != ri).any(axis=1))[0]
    for r in bad:      # f32 near-ties only
        assert np.all(np.abs(s64[r, ti[r, :kk] - 1000] - rv[r]) <= 4 * np.spacing(np.float32(1.0)))
    assert len(bad) <= max(1, B // 50)
    if N < k:
        assert (ti[:, N:] == -1).all()
