"""The opt-in pass 2 on bf16 planes (range_set_pv_mode(RANGE_PV_BF16X3), attend_bf16x3.h): never
the default; its w @ V agrees with the exact float32 kernel to the rounding of the dropped cross
terms (~2^-22 relative to the products), far inside the tolerance the exact kernel itself is
held to against the float64 oracle (2e-5)."""
import numpy as np
import pytest
import torch

from oracle import range_oracle as O
from range_amd import _native
from tools import synth

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to("cuda:0")


def _engine(bank, L=10, H=64):
    w = synth.make_encoder_weights(L, H, 256, 2, 5)
    eng = _native.HipEngine("cuda:0")
    eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC,
                    [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                    [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])
    eng.set_bank(bank.keys, bank.values, bank.xyz)
    return eng


# ragged bank sizes: odd number of 16-row blocks, a last block of one row, fewer rows than a group
@pytest.mark.parametrize("N,B", [(20000, 300), (4113, 70), (1537, 33), (129, 5), (40, 64)])
def test_bf16x3_matches_exact_kernel_and_oracle(N, B):
    bank = O.prep_bank(*synth.make_bank(N, 5))
    eng = _engine(bank)
    assert eng.pv_mode == "exact"
    q = synth.make_queries(B, seed=N + B, lat_max=90.0)
    e64, e32, xq = eng.encode(_dev(q))
    for tau_geo in (40.0, 0.0):
        st = eng.scan_stats(e32, xq, 12.0, tau_geo, keep_logits=True)
        assert eng.kept_queries() == B
        for beta in ((0.5, 0.0, 1.0) if tau_geo > 0 else (1.0,)):
            eng.set_pv_mode("exact")
            ref = eng.attend_kept(0, xq, 12.0, tau_geo, beta, st).cpu().numpy()
            eng.set_pv_mode("bf16x3")
            assert eng.pv_mode == "bf16x3"
            got = eng.attend_kept(0, xq, 12.0, tau_geo, beta, st).cpu().numpy()
            # sum_i |w_i V_i| <= max|V| ~ 4.5 for the synthetic N(0,1) values: 3 * 2^-24 of that
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
            assert np.abs(got - ref).mean() < 1e-7
    # sub-ranges of the kept queries starting on a query tile
    eng.set_pv_mode("bf16x3")
    st = eng.scan_stats(e32, xq, 12.0, 40.0, keep_logits=True)
    whole = eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st)
    for first in range(64, B, 64):
        n = B - first
        part = eng.attend_kept(first, xq[first:first + n], 12.0, 40.0, 0.5, st[first:first + n])
        # (another batch size picks another number of bank splits: same products, another order)
        np.testing.assert_allclose(part.cpu().numpy(), whole[first:first + n].cpu().numpy(), rtol=0, atol=1e-6)
    # against the exact-arithmetic oracle: the tolerance of the exact kernel's own parity test
    e = e64.cpu().numpy()
    ref64 = O.retrieve64(e, q, bank, "RANGE+", 0.5)
    np.testing.assert_allclose(whole.cpu().numpy(), ref64, rtol=0, atol=2e-5)
    # a pass 2 that recomputes its logits stays exact in either mode
    eng.set_pv_mode("exact")
    a = eng.attend(e32, xq, 12.0, 40.0, 0.5, st)
    eng.set_pv_mode("bf16x3")
    assert torch.equal(eng.attend(e32, xq, 12.0, 40.0, 0.5, st), a)


def test_bf16x3_planes_follow_the_bank():
    """range_set_bank after range_set_pv_mode rebuilds the planes; values with a wide dynamic
    range (1e-6 .. 1e3) and exact zeros keep their relative accuracy plane by plane."""
    rng = np.random.default_rng(11)
    locs, vals, keys = synth.make_bank(3000, 9)
    vals = (vals * np.exp(rng.uniform(-14, 7, size=vals.shape))).astype(np.float32)
    vals[::7] = 0.0
    bank = O.prep_bank(locs, vals, keys)
    eng = _native.HipEngine("cuda:0")
    eng.set_pv_mode("bf16x3")                    # before any bank: built by set_bank
    w = synth.make_encoder_weights(10, 64, 256, 2, 5)
    eng.set_encoder(10, 64, 2, 256, _native.SH_ANALYTIC,
                    [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                    [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])
    eng.set_bank(bank.keys, bank.values, bank.xyz)
    q = synth.make_queries(100, seed=3)
    e64, e32, xq = eng.encode(_dev(q))
    st = eng.scan_stats(e32, xq, 12.0, 40.0, keep_logits=True)
    got = eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st).cpu().numpy()
    eng.set_pv_mode("exact")
    ref = eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st).cpu().numpy()
    scale = np.abs(bank.values).max()
    assert np.abs(got - ref).max() < 4e-7 * scale
    # another bank on the same context
    bank2 = O.prep_bank(*synth.make_bank(777, 2))
    eng.set_pv_mode("bf16x3")
    eng.set_bank(bank2.keys, bank2.values, bank2.xyz)
    st = eng.scan_stats(e32, xq, 12.0, 40.0, keep_logits=True)
    got = eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st).cpu().numpy()
    ref64 = O.retrieve64(e64.cpu().numpy(), q, bank2, "RANGE+", 0.5)
    np.testing.assert_allclose(got, ref64, rtol=0, atol=2e-5)
    with pytest.raises(ValueError):
        eng.set_pv_mode("fp8")


def test_load_model_pv_mode(tmp_path):
    """load_model(..., pv_mode='bf16x3') is the opt-in; the default model stays on exact float32
    products.  Both return the reference's (B,1280) float64 ndarray; they agree to 1e-6, the
    embedding half bit for bit; the sweep and a second beta go through the same kernel."""
    from range_amd import load_model
    ck = synth.write_checkpoint(str(tmp_path / "enc.ckpt"), L=10, hidden=64, seed=5)
    db = synth.write_bank(str(tmp_path / "db.npz"), 5000, 3)
    q = torch.from_numpy(synth.make_queries(700, seed=8, lat_max=90.0)).to("cuda:0")
    exact = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db)
    fast = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db, pv_mode="bf16x3")
    assert exact.engine.pv_mode == "exact" and fast.engine.pv_mode == "bf16x3"
    a, b = exact(q), fast(q)
    assert isinstance(b, np.ndarray) and b.dtype == np.float64 and b.shape == (700, 1280)
    assert np.array_equal(a[:, 1024:], b[:, 1024:])
    np.testing.assert_allclose(b[:, :1024], a[:, :1024], rtol=0, atol=1e-6)
    assert not np.array_equal(a[:, :1024], b[:, :1024])          # it IS another arithmetic
    sa, sb = exact.sweep(q, [0.0, 0.3, 1.0]), fast.sweep(q, [0.0, 0.3, 1.0])
    np.testing.assert_allclose(np.asarray(sb), np.asarray(sa), rtol=0, atol=1e-6)
    r = load_model("RANGE", pretrained_path=ck, device="cuda:0", db_path=db, pv_mode="bf16x3")
    np.testing.assert_allclose(r(q)[:, :1024],
                               load_model("RANGE", pretrained_path=ck, device="cuda:0", db_path=db)(q)[:, :1024],
                               rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db, pv_mode="fp8")
