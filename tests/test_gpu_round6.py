"""Round 6 (`-m gpu`): the round-5 verdict's API and edge-case items on the HIP path, each against the
oracle / the reference-generated goldens.

  * ``model(coords, return_topk=k)``: the top-k side channel from the SAME call == ``model.topk`` bit
    for bit == the goldens' ``sem_topk_idx``; host and device contract, every top-k route (one fused
    launch, the GEMM-shaped batch path, chunks);
  * ``model.loc_model`` / ``model.parameters()``: the surface of range/range.py:83-84, 201-203 and
    load_model.py:49-50;
  * the constant-shift softmax (no running maximum; the reference's softmax subtracts it,
    range.py:215, 234) at its extremes: the whole bank at the queries' antipode (geographic terms
    2^-115: ten binades above float32 underflow), at the semantic "antipode", one planted row at
    similarity +1 among rows at -1 - two-pass, one-pass (B <= 32) and the shard merge;
  * degenerate inputs: NaN / infinite coordinates give NaN rows and leave their neighbours alone (as in
    the reference: rows are independent), nothing is reported as a give-up; latitudes beyond +-90 and
    longitudes beyond +-180 continue the way the reference's formulas continue;
  * stream-K pass 2 (the default for banks / shards up to 50 000 rows) against the split scheme and the
    float64 oracle over (N, B) including ragged batches and several bank columns (advisor, round 5).
"""
import os

import numpy as np
import pytest
import torch

from oracle import range_oracle as O
from range_amd import _native
from tools import synth
from range_amd.bank import PreparedBank, prepare_bank
from range_amd.ckpt import EncoderParams

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _params(L, H, layers=2, seed=5, mode="analytic"):
    w = synth.make_encoder_weights(L, H, 256, layers, seed)
    ws = [w[f"layers.{i}.weight"] for i in range(layers)] + [w["last_layer.weight"]]
    bs = [w[f"layers.{i}.bias"] for i in range(layers)] + [w["last_layer.bias"]]
    return w, EncoderParams(L, H, layers, 256, mode, ws, bs)


def _engine(enc, bank=None, row_offset=0, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        eng = _native.HipEngine(DEV)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    eng.set_encoder(enc.legendre_polys, enc.hidden, enc.num_hidden_layers, 256,
                    _native.SH_ANALYTIC if enc.harmonics_calculation == "analytic" else _native.SH_CLOSED_FORM,
                    enc.weights, enc.biases)
    if bank is not None:
        eng.set_bank(bank.keys, bank.values, bank.xyz, row_offset)
    return eng


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


# ----------------------------------------------------------------------------------------------
# forward(coords, return_topk=k)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["e2e_L40_H512_N3000", "e2e_L10_H64_N500"])
def test_forward_return_topk_equals_topk_and_the_reference_golden(tag, tmp_path):
    from range_amd import load_model
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    ck = synth.write_checkpoint(str(tmp_path / "enc.ckpt"), L=int(z["L"]), hidden=int(z["hidden"]),
                                num_hidden_layers=int(z["num_hidden_layers"]), seed=int(z["weight_seed"]))
    db = synth.write_bank(str(tmp_path / "db.npz"), int(z["bank_rows"]), int(z["bank_seed"]))
    q = torch.from_numpy(z["lonlat"]).to(DEV)
    m = load_model("RANGE+", pretrained_path=ck, device=DEV, db_path=db)
    tv, ti = m.topk(q, 16)
    out, fv, fi = m(q, return_topk=16)
    # the reference's contract for the embeddings (host float64 ndarray, range.py:240), its own numbers
    assert isinstance(out, np.ndarray) and out.dtype == np.float64 and np.array_equal(out, m(q))
    np.testing.assert_allclose(out, z["rangeplus_beta0.5"], rtol=0, atol=2e-5)
    assert fv.is_cuda and fi.dtype == torch.int64
    assert torch.equal(fi, ti) and torch.equal(fv, tv)                       # model.topk(), bit for bit
    assert np.array_equal(fi.cpu().numpy(), z["sem_topk_idx"])               # the reference's top-k indices
    dout, dv, di = m(q, return_device=True, return_topk=5)
    assert dout.is_cuda and np.array_equal(dout.cpu().numpy(), out)
    assert torch.equal(di, ti[:, :5]) and torch.equal(dv, tv[:, :5])
    mr = load_model("RANGE", pretrained_path=ck, device=DEV, db_path=db)
    out_r, rv, ri = mr(q, return_topk=16)
    np.testing.assert_allclose(out_r, z["range"], rtol=0, atol=2e-5)
    assert torch.equal(ri, ti)
    with pytest.raises(ValueError, match="return_topk"):
        m(q, return_topk=17)
    with pytest.raises(ValueError, match="return_topk"):
        load_model("SatCLIP", pretrained_path=ck, device=DEV)(q, return_topk=4)


@pytest.mark.parametrize("N,B,chunk", [(700, 5, None), (20_000, 200, None), (20_000, 700, None), (5000, 1000, 256), (3000, 40, 16)])
def test_forward_return_topk_on_every_route(N, B, chunk, tmp_path):
    """5 queries: one-pass forward + the fused one-launch scan; 200: two-pass + fused scan; 700: the
    GEMM-shaped batch top-k; chunks: the e-hat of EACH chunk's forward serves that chunk's scan."""
    from range_amd import load_model
    L, H = 10, 64
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=5)
    db = synth.write_bank(str(tmp_path / "db.npz"), N, 77)
    m = load_model("RANGE+", pretrained_path=ck, device=DEV, db_path=db, beta=0.25)
    if chunk:
        m.chunk_size = chunk
        m.topk_stream_max = chunk
    qn = synth.make_queries(B, seed=9, lat_max=90.0)
    q = _dev(qn)
    plain = m(q)
    tv, ti = m.topk(q, 16)
    out, fv, fi = m(q, return_topk=16)
    assert np.array_equal(out, plain) and torch.equal(fi, ti) and torch.equal(fv, tv)
    locs, vals, keys = synth.make_bank(N, 77)
    s, _ = O.logits64(out[:, 1024:], qn, O.prep_bank(locs, vals, keys))
    rv, ri = O.topk64(s, 16)
    assert (fi.cpu().numpy() != ri).any(axis=1).sum() <= 1                    # (4-ulp ties aside)
    np.testing.assert_allclose(fv.cpu().numpy(), rv, rtol=0, atol=3e-7)
    # the workspace belongs to the LAST forward: a stale request is refused, not answered from old rows
    m.engine.forward(q[: max(1, B // 2)].contiguous(), _native.MODEL_RANGE_PLUS, 0.25)
    with pytest.raises(_native.RangeNativeError, match="range_topk_last"):
        m.engine.topk_last(B, 16)


# ----------------------------------------------------------------------------------------------
# loc_model / parameters: range/range.py:83-84, 201-203; load_model.py:49-50
# ----------------------------------------------------------------------------------------------
def test_loc_model_and_parameters_surface(tmp_path):
    from range_amd import load_model
    L, H = 10, 64
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=5)
    db = synth.write_bank(str(tmp_path / "db.npz"), 300, 3)
    w = synth.make_encoder_weights(L, H, 256, 2, 5)
    qn = synth.make_queries(70, seed=2)
    q = _dev(qn)
    for name in ("RANGE+", "RANGE", "SatCLIP"):
        m = load_model(name, pretrained_path=ck, device=DEV, **({"db_path": db} if "RANGE" in name else {}))
        # range.py:210 / :244: self.loc_model(coords) -> the RAW (B,256) float64 embedding, on the device
        raw = m.loc_model(q)
        assert raw.is_cuda and raw.dtype == torch.float64 and tuple(raw.shape) == (70, 256)
        np.testing.assert_allclose(raw.cpu().numpy(), O.siren_forward(O.sh_features(qn, L), w), rtol=0, atol=1e-10)
        if name == "SatCLIP":
            assert torch.equal(raw, m(q))
        else:
            e = raw / raw.norm(p=2, dim=-1, keepdim=True)                     # range.py:212
            np.testing.assert_allclose(m(q)[:, 1024:], e.cpu().numpy(), rtol=0, atol=1e-15)
        # range.py:201-203: eval mode, every parameter frozen; load_model.py:49-50: .eval().to(device)
        ps = list(m.parameters())
        assert ps and all(not p.requires_grad for p in ps) and not m.training and not m.loc_model.training
        assert next(m.parameters()).device == torch.device(DEV) and next(m.parameters()).dtype == torch.float64
        assert m.to(DEV) is m and m.eval() is m
        # the SirenNet's parameters under the reference's names (the checkpoint's model.location.* keys)
        sd = m.loc_model.state_dict()
        assert sorted(sd) == sorted(f"nnet.{k}" for k in w)
        for k, v in w.items():
            assert np.array_equal(sd[f"nnet.{k}"].cpu().numpy(), v)
        assert list(m.state_dict()) == [f"loc_model.{k}" for k in sd]
    # the training-free encoders: a parameter-free loc_model there too (DummyLocationEncoder / Wrap())
    for name, d in (("Direct", 2), ("Cartesian_3D", 3), ("Wrap", 4)):
        m = load_model(name, pretrained_path=ck, device=DEV)
        assert list(m.parameters()) == [] and m.location_feature_dim == d
        if name == "Wrap":
            assert torch.equal(m.loc_model(q), m(q))
        else:
            assert m.loc_model(q) is q


# ----------------------------------------------------------------------------------------------
# constant-shift softmax at its extremes
# ----------------------------------------------------------------------------------------------
def _forward_all_routes(enc, obank, qn, e_ref_check=True):
    """RANGE+ (beta 0 / .5 / 1) and RANGE through range_forward for this batch, against the float64
    oracle (2e-5) and the reference's float32 op order (1e-4); returns the beta=0.5 device result."""
    bank = PreparedBank(obank.keys, obank.values, obank.xyz)
    eng = _engine(enc, bank)
    x = _dev(qn)
    keep = None
    for model, name, betas in ((_native.MODEL_RANGE_PLUS, "RANGE+", (0.0, 0.5, 1.0)), (_native.MODEL_RANGE, "RANGE", (1.0,))):
        for beta in betas:
            out = eng.forward(x, model, beta).cpu().numpy()
            assert np.isfinite(out).all()
            e = out[:, 1024:]
            np.testing.assert_allclose(out[:, :1024], O.retrieve64(e, qn, obank, name, beta), rtol=0, atol=2e-5,
                                       err_msg=f"{name} beta={beta} B={len(qn)} N={obank.keys.shape[0]}")
            np.testing.assert_allclose(out, O.retrieve(e, qn, obank, name, beta), rtol=0, atol=1e-4)
            if name == "RANGE+" and beta == 0.5:
                keep = out
    eng.check_async_error()
    return keep


def _antipodal_case(N, B, L=10, H=64, seed=0):
    """Bank locations inside a 1-degree cap around (30 E, 20 N); queries within 0.25 degrees of its
    antipode (150 W, 20 S): every geographic logit is within 1e-4 of -1, every term of the geographic
    softmax 2^(-2 * 40 * log2 e) = 2^-115.  The keys sit at the semantic antipode of query 0 (-e-hat_0 +
    1 % noise): that query's semantic logits are all ~ -1 too."""
    rng = np.random.default_rng(seed)
    w, enc = _params(L, H)
    qn = np.stack([-150.0 + rng.uniform(-0.25, 0.25, B), -20.0 + rng.uniform(-0.25, 0.25, B)], axis=1)
    e0 = O.encode(qn[:1], w, L)[0]
    locs = np.stack([30.0 + rng.uniform(-0.5, 0.5, N), 20.0 + rng.uniform(-0.5, 0.5, N)], axis=1)
    keys = -e0[None, :] + 0.01 * rng.standard_normal((N, 256))
    vals = rng.standard_normal((N, 1024)).astype(np.float32)
    return enc, O.prep_bank(locs, vals, keys), qn, w


@pytest.mark.parametrize("N", [1, 9, 16, 10_000])
@pytest.mark.parametrize("B", [7, 20, 40])
def test_softmax_with_the_whole_bank_at_the_antipode(N, B):
    """B = 7 / 20: ONE pass over the bank (attend_small.h, one / two query tiles per workgroup); B = 40:
    the two-pass kernels (pass 1 statistics, pass 2 on kept logits)."""
    enc, obank, qn, w = _antipodal_case(N, B)
    g = O.logits64(O.encode(qn, w, 10), qn, obank)[1]
    assert g.max() < -0.9997                      # the premise: every geographic similarity ~ -1
    s0 = O.logits64(O.encode(qn, w, 10), qn, obank)[0][0]
    assert s0.max() < -0.98                       # ... and query 0's semantic ones
    _forward_all_routes(enc, obank, qn)


@pytest.mark.parametrize("N", [9, 16, 10_000])
def test_antipodal_bank_through_the_shard_merge(N):
    """Two engines holding the halves of the antipodal bank, statistics merged, partials added: the
    sums l of the shards are ~N/2 * 2^-115 each and must add without a rescale."""
    B = 40
    enc, obank, qn, w = _antipodal_case(N, B)
    bank = PreparedBank(obank.keys, obank.values, obank.xyz)
    cut = N // 2
    full, a, b = _engine(enc, bank), _engine(enc, bank.rows(0, cut), 0), _engine(enc, bank.rows(cut, N), cut)
    e64, e32, xq = full.encode(_dev(qn))
    for tau_sem, tau_geo, beta, name in ((12.0, 40.0, 0.5, "RANGE+"), (12.0, 40.0, 0.0, "RANGE+"), (15.0, 0.0, 1.0, "RANGE")):
        st = full.merge_stats(torch.stack([a.scan_stats(e32, xq, tau_sem, tau_geo), b.scan_stats(e32, xq, tau_sem, tau_geo)]))
        two = full.finalize(torch.stack([a.attend(e32, xq, tau_sem, tau_geo, beta, st),
                                         b.attend(e32, xq, tau_sem, tau_geo, beta, st)]), e64).cpu().numpy()
        one = full.forward(_dev(qn), _native.MODEL_RANGE_PLUS if tau_geo > 0 else _native.MODEL_RANGE, beta).cpu().numpy()
        np.testing.assert_allclose(two, one, rtol=0, atol=2e-6)
        np.testing.assert_allclose(two[:, :1024], O.retrieve64(two[:, 1024:], qn, obank, name, beta), rtol=0, atol=2e-5)


@pytest.mark.parametrize("B", [5, 40])
def test_one_planted_row_at_similarity_one_among_rows_at_minus_one(B):
    """Query 0: one bank row carries its own e-hat and its own location, every other row sits at both
    antipodes.  The softmax is that one row (the others weigh e^-24 / e^-80 each): the output must be
    the planted row's values - and the row's logit 2^0 must coexist with terms of 2^-115 in one sum."""
    N, L, H = 10_000, 10, 64
    rng = np.random.default_rng(3)
    w, enc = _params(L, H)
    qn = np.stack([-150.0 + rng.uniform(-0.25, 0.25, B), -20.0 + rng.uniform(-0.25, 0.25, B)], axis=1)
    e0 = O.encode(qn[:1], w, L)[0]
    locs = np.stack([30.0 + rng.uniform(-0.5, 0.5, N), 20.0 + rng.uniform(-0.5, 0.5, N)], axis=1)
    keys = -e0[None, :] + 0.01 * rng.standard_normal((N, 256))
    r = 6151
    locs[r] = qn[0]
    keys[r] = e0
    vals = rng.standard_normal((N, 1024)).astype(np.float32)
    obank = O.prep_bank(locs, vals, keys)
    out = _forward_all_routes(enc, obank, qn)
    np.testing.assert_allclose(out[0, :1024], vals[r], rtol=0, atol=2e-5)
    eng = _engine(enc, PreparedBank(obank.keys, obank.values, obank.xyz))
    _, ti = eng.topk_stream(eng.encode(_dev(qn))[1], 4)
    assert int(ti[0, 0]) == r


# ----------------------------------------------------------------------------------------------
# degenerate coordinates
# ----------------------------------------------------------------------------------------------
BAD = np.array([[np.nan, 10.0], [10.0, np.nan], [np.inf, 0.0], [0.0, -np.inf], [np.nan, np.nan]])


@pytest.mark.parametrize("B", [8, 40, 600, 3000])
def test_nan_and_infinite_coordinates_give_nan_rows_and_nothing_else(B, tmp_path):
    """The reference: a NaN / infinite coordinate makes that query's features NaN and with them its row
    - the rows of a batch are independent (range.py:206-240).  Same here, on every route (one pass, two
    passes, the one-launch encoder, the large-batch encoder), and it is NOT a give-up: nothing raises."""
    from range_amd import load_model
    L, H, N = 10, 64, 4000
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=5)
    db = synth.write_bank(str(tmp_path / "db.npz"), N, 77)
    m = load_model("RANGE+", pretrained_path=ck, device=DEV, db_path=db)
    good = synth.make_queries(B, seed=4, lat_max=90.0)
    qn = good.copy()
    rows = np.array([0, 1, B // 2, B - 2, B - 1])[: len(BAD)]
    qn[rows] = BAD
    out = m(torch.from_numpy(qn))
    m.engine.check_async_error()
    isnan = np.isnan(out).all(axis=1)
    assert np.array_equal(np.flatnonzero(isnan), np.unique(rows)) and not np.isnan(out[~isnan]).any()
    ref = m(torch.from_numpy(good))
    ok = np.setdiff1d(np.arange(B), rows)
    np.testing.assert_allclose(out[ok], ref[ok], rtol=0, atol=1e-12)           # the neighbours: untouched
    # the oracle (the reference's op order) agrees about which rows are NaN
    w = synth.make_encoder_weights(L, H, 256, 2, 5)
    locs, vals, keys = synth.make_bank(N, 77)
    with np.errstate(invalid="ignore"):
        oref = O.forward(qn[:64], w, L, O.prep_bank(locs, vals, keys), "RANGE+", 0.5)
    assert np.array_equal(np.isnan(oref).all(axis=1), isnan[:64])
    # the top-k of the good rows does not see its NaN neighbours either
    tv, ti = m.topk(torch.from_numpy(qn), 8)
    rv, ri = m.topk(torch.from_numpy(good), 8)
    assert torch.equal(ti[torch.from_numpy(ok)], ri[torch.from_numpy(ok)])
    m.engine.check_async_error()


def test_latitudes_beyond_the_poles_and_longitudes_beyond_the_date_line_continue_like_the_reference():
    """|lat| > 90, |lon| > 180: the reference does not validate - theta = rad(lat + 90) leaves [0, pi],
    its polynomials in cos(theta) and |sin(theta)|^m go on (spherical_harmonics.py:31-42); the
    geographic head's rad_to_cart is plain trigonometry (utils.py:11-16).  Both SH evaluations follow:
    the 'reference' mode against the oracle's reference-shaped evaluation (bitwise the reference's
    features), the recurrence against the oracle's."""
    from range_amd import sh_table
    L, H = 40, 64
    w, enc = _params(L, H, seed=1234)
    rng = np.random.default_rng(5)
    n = 120
    lat = np.concatenate([rng.uniform(135, 150, n // 2), -rng.uniform(135, 150, n // 2)])     # cos(theta) as for |lat| in (30, 45): well-conditioned
    lon = rng.uniform(-400, 400, n)
    qn = np.stack([lon, lat], axis=1)
    eng = _native.HipEngine(DEV)
    eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, enc.weights, enc.biases, sh_table=sh_table.generate_table(L))
    e64, e32, xq = eng.encode(_dev(qn))
    ref = O.encode(qn, w, L, features=O.sh_features_faithful(qn, O.load_ylm_table(), L))
    np.testing.assert_allclose(e64.cpu().numpy(), ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(xq.cpu().numpy()[:, :3], O.query_xyz(qn), rtol=0, atol=2.4e-7)
    for mode in ("analytic", "closed-form"):
        w2, enc2 = _params(L, H, seed=1234, mode=mode)
        ex = _engine(enc2).encode(_dev(qn))[0].cpu().numpy()
        np.testing.assert_allclose(ex, O.encode(qn, w2, L, mode), rtol=0, atol=2e-12)
    # in the well-conditioned band the two evaluations agree, beyond the poles as inside them
    np.testing.assert_allclose(e64.cpu().numpy(), O.encode(qn, w, L), rtol=0, atol=1e-5)


# ----------------------------------------------------------------------------------------------
# stream-K pass 2 against the split scheme (advisor, round 5)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [12_500, 20_011, 33_000, 50_000])
def test_streamk_pass2_against_the_split_scheme_and_the_oracle(N):
    """The default pass 2 for banks / shards of up to 50 000 rows is a persistent stream-K walk (one
    workgroup per CU over (query tile, bank block) units; 1 .. 4 bank columns of <= 16 384 rows); a query
    tile's cut points depend on its INDEX, so the float32 rounding of a query depends on its position in
    the batch (documented in range_hip.h).  Against RANGE_P2_STREAMK=0 (one workgroup per (split, tile):
    position-independent) within float32 summation-order rounding, and against the float64 oracle, for
    the kept and the recompute kernel, ragged and full batches."""
    L, H = 10, 64
    w, enc = _params(L, H)
    locs, vals, keys = synth.make_bank(N, 31)
    bank, obank = prepare_bank(locs, vals, keys), O.prep_bank(locs, vals, keys)
    sk = _engine(enc, bank)
    sp = _engine(enc, bank, env={"RANGE_P2_STREAMK": "0"})
    rc = _engine(enc, bank, env={"RANGE_KEEP_LOGITS": "0"})
    rng = np.random.default_rng(N)
    for B in (33, 1250, 10_000, 16_384):
        qn = synth.make_queries(B, seed=B, lat_max=90.0)
        x = _dev(qn)
        for beta in (0.5,) if B > 5000 else (0.0, 0.5, 1.0):
            a = sk.forward(x, _native.MODEL_RANGE_PLUS, beta)
            assert sk.kept_queries() == B
            b = sp.forward(x, _native.MODEL_RANGE_PLUS, beta)
            c = rc.forward(x, _native.MODEL_RANGE_PLUS, beta)
            assert rc.kept_queries() == 0
            assert float((a - b).abs().max()) < 2e-6 and float((a - c).abs().max()) < 2e-6
            assert torch.equal(a[:, 1024:], b[:, 1024:])
            idx = np.sort(rng.choice(B, min(B, 24), replace=False))
            got = a[torch.from_numpy(idx).to(DEV)].cpu().numpy()
            np.testing.assert_allclose(got[:, :1024], O.retrieve64(got[:, 1024:], qn[idx], obank, "RANGE+", beta), rtol=0, atol=2e-5)
        # the same queries at another position of an equal-sized batch: equal within rounding, e-hat bit for bit
        if B >= 1250:
            perm = torch.from_numpy(rng.permutation(B)).to(DEV)
            ap = sk.forward(x[perm].contiguous(), _native.MODEL_RANGE_PLUS, 0.5)
            a = sk.forward(x, _native.MODEL_RANGE_PLUS, 0.5)
            assert float((ap - a[perm]).abs().max()) < 2e-6


# ----------------------------------------------------------------------------------------------
# two rank processes (gloo, sharing the GPU): return_topk over shards; a peer's give-up is learnt from
# its FLAG on every rank; NaN coordinates are not a give-up (advisor, round 5)
# ----------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sharded_rank(rank, world, port, ck, db, tmp, ret):
    import torch.distributed as dist
    from argparse import Namespace
    from range_amd import load_model
    from range_amd.dist import init_from_env
    from range_amd.save import save_embeddings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    init_from_env("gloo", timeout_s=60)
    try:
        m = load_model("RANGE+", pretrained_path=ck, device=DEV, db_path=db, beta=0.5, shards=world)
        assert not next(m.parameters()).requires_grad and next(m.parameters()).device == torch.device(DEV)
        qn = synth.make_queries(1501, seed=21, lat_max=90.0)
        q = torch.from_numpy(qn)
        # ---- return_topk over the shards: the full batch on every rank, == the one-GPU model (rank 0 checks)
        out, tv, ti = m(q, return_topk=16)
        tv2, ti2 = m.topk(q, 16)
        assert isinstance(out, np.ndarray) and out.shape == (1501, 1280) and torch.equal(ti, ti2) and torch.equal(tv, tv2)
        assert np.array_equal(out, m(q))
        lo, hi = (1501 * rank) // world, (1501 * (rank + 1)) // world
        lout, lv, li = m(q[lo:hi], local=True, return_topk=16)
        assert np.array_equal(lout, out[lo:hi]) and torch.equal(li, ti[lo:hi]) and torch.equal(lv, tv[lo:hi])
        if rank == 0:
            m1 = load_model("RANGE+", pretrained_path=ck, device=DEV, db_path=db, beta=0.5)
            o1, v1, i1 = m1(q, return_topk=16)
            assert torch.equal(i1, ti) and torch.equal(v1, tv)
            np.testing.assert_allclose(out, o1, rtol=0, atol=2e-6)
            del m1
        dist.barrier()
        # ---- NaN coordinates: NaN rows, the other rows untouched, NO refusal - drop-in call and batch driver
        bad = qn.copy()
        bad[[3, 700, 1500]] = [[np.nan, 1.0], [2.0, np.inf], [np.nan, np.nan]]
        ob = m(torch.from_numpy(bad))
        nanrows = np.flatnonzero(np.isnan(ob).all(axis=1))
        assert list(nanrows) == [3, 700, 1500]
        ok = np.setdiff1d(np.arange(1501), nanrows)
        np.testing.assert_allclose(ob[ok], out[ok], rtol=0, atol=2e-6)

        def loader(coords):
            for i in range(0, len(coords), 600):
                c = coords[i:i + 600]
                yield torch.from_numpy(c), torch.arange(len(c), dtype=torch.float32)
        a = Namespace(embeddings_dir=os.path.join(tmp, "emb"), location_model_name="RANGE+", task_name="t")
        save_embeddings(a, loader(bad), loader(qn[:100]), m)
        if rank == 0:
            z = np.load(os.path.join(tmp, "emb", "RANGE+", "t_train.npz"))
            assert list(np.flatnonzero(np.isnan(z["embeddings"]).all(axis=1))) == [3, 700, 1500]
            np.testing.assert_allclose(z["embeddings"][ok], out[ok], rtol=0, atol=2e-6)
        dist.barrier()
        # ---- a persistent launch of rank 1 gives up: EVERY rank refuses the call, from the flag
        small = torch.from_numpy(synth.make_queries(64, seed=5))         # 32 own queries: the one-launch encoder
        m(small)
        if rank == 1:
            m.engine.debug_fail_next_persistent_launch()
        with pytest.raises((RuntimeError, _native.RangeNativeError), match="gave up"):
            m(small)
        dist.barrier()
        again = m(small)                                                  # re-issued: the fall-back path, finite rows
        assert np.isfinite(again).all()
        # ... and in the batch driver: every rank raises in the same batch (none is left in a collective)
        e2 = load_model("RANGE+", pretrained_path=ck, device=DEV, db_path=db, beta=0.5, shards=world)
        if rank == 1:
            e2.engine.debug_fail_next_persistent_launch()
        b = Namespace(embeddings_dir=os.path.join(tmp, "emb2"), location_model_name="RANGE+", task_name="t")
        with pytest.raises((RuntimeError, _native.RangeNativeError), match="gave up"):
            save_embeddings(b, loader(qn[:64]), loader(qn[:64]), e2)
        dist.barrier()
        ret[rank] = "ok"
    except BaseException as ex:  # noqa: BLE001
        import traceback
        ret[rank] = f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


def test_sharded_return_topk_flags_and_nan_coordinates_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    L, H, N = 20, 256, 5001          # (a shape whose small batches take the one-launch - persistent - encoder)
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=5)
    db = synth.write_bank(str(tmp_path / "db.npz"), N, 77)
    ret = mp.Manager().dict()
    mp.spawn(_sharded_rank, args=(2, _free_port(), ck, db, str(tmp_path), ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)
