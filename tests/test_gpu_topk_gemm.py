"""Batch-scale top-k (range_amd/csrc/topk_gemm.h: batches beyond 256 queries): the GEMM-shaped,
list-free path must return the float32 scan's values and indices BIT FOR BIT - the same contract the
streaming scan has (tests/test_gpu_round2.py) - and the float64 oracle's indices (float32 near-ties
aside): ragged banks and batches, every k, a crowd of rows inside the bf16 error band around the 16th
place, un-normalised keys and queries, a last tile of pad rows that would outrank every real row, and
candidate lists that overflow (answered by brute force)."""
import numpy as np
import pytest
import torch

from oracle import range_oracle as O
from range_amd import _native

pytestmark = pytest.mark.gpu


def _unit(rng, n, d=256):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _engines(monkeypatch, keys, row_offset=0):
    """(default engine: GEMM path beyond 256 queries; engine with the streaming scan at every size)"""
    gemm = _native.HipEngine("cuda:0")
    monkeypatch.setenv("RANGE_TOPK_GEMM", "0")
    stream = _native.HipEngine("cuda:0")
    monkeypatch.delenv("RANGE_TOPK_GEMM")
    for e in (gemm, stream):
        e.set_keys(keys, row_offset)
    return gemm, stream


def _check_oracle(tv, ti, qs, keys, k, row_offset=0, atol=4e-7):
    s64 = qs.astype(np.float64) @ keys.astype(np.float64).T
    kk = min(k, keys.shape[0])
    rv, ri = O.topk64(s64, kk)
    tv, ti = tv.cpu().numpy(), ti.cpu().numpy()
    scale = max(1.0, float(np.abs(rv).max()))
    np.testing.assert_allclose(tv[:, :kk], rv, rtol=0, atol=atol * scale)
    bad = np.nonzero((ti[:, :kk] - row_offset != ri).any(axis=1))[0]
    for r in bad:      # float32 near-ties only
        assert np.all(np.abs(s64[r, ti[r, :kk] - row_offset] - rv[r]) <= max(4 * np.spacing(np.float32(scale)), atol))
    assert len(bad) <= max(1, qs.shape[0] // 50)


@pytest.mark.parametrize("N,B,k", [(1024, 257, 16), (1029, 300, 3), (4097, 513, 16), (20011, 1000, 7),
                                   (70_000, 2049, 16), (100_003, 4096, 16), (1500, 3000, 1)])
def test_gemm_topk_equals_the_streaming_scan_and_the_oracle(N, B, k, monkeypatch):
    rng = np.random.default_rng(N + B)
    keys, qs = _unit(rng, N), _unit(rng, B)
    gemm, stream = _engines(monkeypatch, keys, row_offset=1000)
    e = torch.from_numpy(qs).cuda()
    av, ai = gemm.topk_stream(e, k)
    bv, bi = stream.topk_stream(e, k)
    assert torch.equal(ai, bi) and torch.equal(av, bv)
    _check_oracle(av, ai, qs, keys, k, row_offset=1000)
    assert gemm.topk_stream_exact_count() == 0


def test_gemm_topk_crowds_norms_and_pad_rows(monkeypatch):
    rng = np.random.default_rng(5)
    N, B, d = 30011, 600, 256
    keys, q = _unit(rng, N), _unit(rng, B)
    # a crowd around the 16th place of queries 0..7: 60 rows whose similarity to query b is
    # 0.5 + i 2e-6 - far inside the bf16 band (8e-3), above float32 resolution
    for b in range(8):
        rows = rng.choice(N, 60, replace=False)
        for i, r in enumerate(rows):
            u = rng.standard_normal(d)
            u -= u.dot(q[b].astype(np.float64)) * q[b]
            u /= np.linalg.norm(u)
            s = 0.5 + i * 2e-6
            keys[r] = (s * q[b] + np.sqrt(1 - s * s) * u).astype(np.float32)
    gemm, stream = _engines(monkeypatch, keys)
    for qs in (q, q[:300], q * np.float32(3.7)):
        for k in (16, 7, 1):
            e = torch.from_numpy(np.ascontiguousarray(qs)).cuda()
            av, ai = gemm.topk_stream(e, k)
            bv, bi = stream.topk_stream(e, k)
            assert torch.equal(ai, bi) and torch.equal(av, bv)
    assert gemm.topk_stream_exact_count() == 0
    _check_oracle(*gemm.topk_stream(torch.from_numpy(q[8:]).cuda(), 16), q[8:], keys, 16)
    # keys of other norms: rows scaled by 0.25 .. 4 (the error bound scales with the largest norm)
    scale = rng.uniform(0.25, 4.0, size=(N, 1)).astype(np.float32)
    g2, s2 = _engines(monkeypatch, keys * scale)
    e = torch.from_numpy(q).cuda()
    av, ai = g2.topk_stream(e, 16)
    bv, bi = s2.topk_stream(e, 16)
    assert torch.equal(ai, bi) and torch.equal(av, bv)
    # every real row has a NEGATIVE similarity to every query; the last tile's 13 pad rows (zero keys:
    # similarity 0) must not be found, nor shift a threshold
    Np = 4099
    base = _unit(rng, 1)[0]
    keysn = -(base[None, :] + 0.05 * rng.standard_normal((Np, d))).astype(np.float32)
    qn = (base[None, :] + 0.05 * rng.standard_normal((400, d))).astype(np.float32)
    g3, s3 = _engines(monkeypatch, keysn)
    e = torch.from_numpy(qn).cuda()
    av, ai = g3.topk_stream(e, 16)
    bv, bi = s3.topk_stream(e, 16)
    assert torch.equal(ai, bi) and torch.equal(av, bv)
    assert float(av.max()) < 0.0 and int(ai.max()) < Np and int(ai.min()) >= 0
    # (256 products of one sign, vectors of norm 1.3: the float32 chain itself is 8e-7 from float64 here)
    _check_oracle(av, ai, qn, keysn, 16, atol=2e-6)


def test_gemm_topk_overflowing_lists_fall_back_to_brute_force(monkeypatch):
    """5 000 copies of one row and 5 000 near-copies: every list of the affected queries overflows ->
    brute force over all rows, ties in row order; queries far from the copies stay on the fast path."""
    rng = np.random.default_rng(9)
    N, d = 12000, 256
    keys = _unit(rng, N)
    hot = _unit(rng, 1)[0]
    keys[1000:6000] = hot
    near = hot[None, :] + 1e-4 * rng.standard_normal((5000, d)).astype(np.float32)
    keys[6000:11000] = near / np.linalg.norm(near, axis=1, keepdims=True)
    q = _unit(rng, 300)
    q[:5] = hot + 0.01 * rng.standard_normal((5, d)).astype(np.float32)
    gemm, stream = _engines(monkeypatch, keys)
    e = torch.from_numpy(q).cuda()
    av, ai = gemm.topk_stream(e, 16)
    bv, bi = stream.topk_stream(e, 16)
    assert torch.equal(ai, bi) and torch.equal(av, bv)
    assert gemm.topk_stream_exact_count() >= 5
    s64 = q[:5].astype(np.float64) @ keys.astype(np.float64).T
    assert np.all(np.take_along_axis(s64, ai[:5].cpu().numpy(), 1) >= np.sort(s64, axis=1)[:, -16][:, None] - 1e-6)


@pytest.mark.parametrize("key_scale", [1e-20, 1.0, 3.0e3])
def test_gemm_topk_operand_scales(key_scale, monkeypatch):
    """The fp16 operands of the GEMM passes are the keys and queries times powers of two chosen from
    their magnitudes (topk_gemm.h): banks and queries far from unit scale, queries with one dominant
    element (the others sink towards fp16's subnormals after scaling), rows of very different norms and
    an all-zero query (every row ties: lists overflow, brute force) must still give the float32 scan's
    values and indices bit for bit."""
    rng = np.random.default_rng(77)
    N, B = 20_011, 400
    keys = _unit(rng, N) * np.float32(key_scale)
    keys[::7] *= np.float32(1e-3)                       # rows of very different norms
    keys[5::101] *= np.float32(0.25)
    q = _unit(rng, B)
    q[0:40] *= np.float32(1e-30)
    q[40:80] *= np.float32(1e20)
    spike = rng.integers(0, 256, 40)
    q[80:120] *= np.float32(1e-6)
    q[np.arange(80, 120), spike] = 1.0                  # one element a million times the rest
    q[120:160, 1:] = 0.0                                # a single non-zero element
    q[160] = 0.0                                        # the zero query
    gemm, stream = _engines(monkeypatch, keys)
    e = torch.from_numpy(np.ascontiguousarray(q)).cuda()
    for k in (16, 5):
        av, ai = gemm.topk_stream(e, k)
        bv, bi = stream.topk_stream(e, k)
        assert torch.equal(ai, bi) and torch.equal(av, bv), k
    assert bool(torch.isfinite(av).all())
    s64 = q.astype(np.float64) @ keys.astype(np.float64).T
    rv, ri = O.topk64(s64, 16)
    av, ai = gemm.topk_stream(e, 16)
    got = ai.cpu().numpy()[161:]
    assert int((got != ri[161:]).any(axis=1).sum()) <= 3      # (float32 near-ties)
    # the VALUES are float32 similarities at every magnitude (round 5: the streaming scan handed out its
    # bf16 approximations for queries whose squared norm underflows float32 - and agreed with itself)
    scale = (np.linalg.norm(q.astype(np.float64), axis=1) * np.linalg.norm(keys.astype(np.float64), axis=1).max())[:, None]
    ok = scale[:, 0] >= 1e-33          # (where float32 can hold the products at all)
    assert ok.sum() >= 359 and float((np.abs(av.cpu().numpy() - rv)[ok] / scale[ok]).max()) < 2e-6
    # ... and from the one-launch regime of the streaming scan (<= 256 queries) as well
    for lo in (0, 40, 80):
        sv, si = gemm.topk_stream(e[lo:lo + 40].contiguous(), 16)
        if ok[lo:lo + 40].all():
            assert float((np.abs(sv.cpu().numpy() - rv[lo:lo + 40]) / scale[lo:lo + 40]).max()) < 2e-6
            assert int((si.cpu().numpy() != ri[lo:lo + 40]).any(axis=1).sum()) <= 2
