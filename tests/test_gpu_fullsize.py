"""BASELINE.json's single-GPU configurations at FULL size, through the drop-in API
(``load_model(...)(q, return_device=True)``), i.e. through the kernels and the launch geometry the
benchmark times (pass 1 keeping its logits, pass 2 on the kept logits, split slabs summed by the
finalize kernel):

  C2  RANGE+ beta=0.5, range_db_med   (N =  50 000), 10 000 queries, top-16 indices bit-exact
  C3  RANGE+ beta=0.5, range_db_large (N = 100 000), 100 000 queries in chunks of 16 384

A dense oracle at these sizes would take minutes, so every row is checked through size-independent
properties (planted constant value columns are reproduced - the softmax weights of each row sum
to one over all N rows; every output lies inside the range of the bank values; e-hat rows are unit
vectors and equal the separately encoded ones) and a 256-query sample is checked against the
float64 oracle (2e-5), the reference's float32 op order (1e-4, the north-star tolerance) and the
oracle's top-16.  Plus: ``bench.py --gpus 2`` started from a plain shell (two gloo ranks sharing the
one GPU of the test box) as a subprocess.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import range_oracle as O
from range_amd import load_model
from tools import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, H, SEED = 40, 512, 1234


def _model(tmp_path, N):
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=SEED)
    locs, vals, keys = synth.make_bank(N, 2024)
    vals = vals.copy()
    vals[:, 0] = 1.0            # constant columns: reproduced iff the weights of a row sum to one
    vals[:, 1] = -2.5
    db = str(tmp_path / f"db{N}.npz")
    np.savez(db, locs=locs, image_embeddings=vals, satclip_embeddings=keys)   # generate_db.py:212-214
    m = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db, beta=0.5)
    return m, O.prep_bank(locs, vals, keys), synth.make_encoder_weights(L, H, 256, 2, SEED)


def _check_all_rows(out, m, x, vmin, vmax, mean_tol=3e-6):
    assert out.shape == (x.shape[0], 1280) and out.dtype == torch.float64 and out.is_cuda
    assert bool(torch.isfinite(out).all())
    # float32 sums of N weights on both sides (pass 1's l, pass 2's MFMA accumulation): the
    # worst row of 10^4..10^5 queries sits at ~40 ulp, the mean at ~7 ulp (sequential f32 sums of ~7700 terms per split)
    assert float((out[:, 0] - 1.0).abs().max()) < 1e-5
    assert float((out[:, 1] + 2.5).abs().max()) < 2.5e-5
    assert float((out[:, 0] - 1.0).abs().mean()) < mean_tol
    assert float(out[:, :1024].max()) <= vmax and float(out[:, :1024].min()) >= vmin
    assert float((out[:, 1024:].norm(dim=1) - 1.0).abs().max()) < 1e-12
    # the e-hat half is the encoder's output, whatever chunk / workgroup the query was in (the
    # float64 summation order of a workgroup depends on its position, not the values beyond ~1e-15)
    for lo in (0, x.shape[0] - 4096):
        e64, _, _ = m.engine.encode(x[lo:lo + 4096])
        assert float((out[lo:lo + 4096, 1024:] - e64).abs().max()) < 1e-13


def _check_sample(out, m, q, x, obank, w, n=256):
    idx = np.sort(np.random.default_rng(1).choice(q.shape[0], n, replace=False))
    got = out[torch.from_numpy(idx).to(out.device)].cpu().numpy()
    qs = q[idx]
    # e-hat: the model runs the reference's generated SH polynomials (load_model's default for an
    # analytic checkpoint); they ARE the exact basis to 1e-9 inside |lat| <= 30 and drift from it
    # towards the poles exactly as the reference does (tests/test_gpu_parity.py::
    # test_encoder_reference_mode_over_all_latitudes).  The retrieval is checked given that e-hat.
    e = got[:, 1024:]
    low = np.abs(qs[:, 1]) <= 30
    np.testing.assert_allclose(e[low], O.encode(qs[low], w, L), rtol=0, atol=5e-9)
    assert np.abs(e - O.encode(qs, w, L)).max() < 5e-2
    np.testing.assert_allclose(got[:, :1024], O.retrieve64(e, qs, obank, "RANGE+", 0.5), rtol=0, atol=2e-5)
    np.testing.assert_allclose(got, O.retrieve(e, qs, obank, "RANGE+", 0.5), rtol=0, atol=1e-4)
    # top-16 side channel of the same queries: indices equal to the float64 oracle's
    tv, ti = m.topk(x[torch.from_numpy(idx).to(x.device)], 16)
    s64, _ = O.logits64(e, qs, obank)
    rv, ri = O.topk64(s64, 16)
    np.testing.assert_allclose(tv.cpu().numpy(), rv, rtol=0, atol=3e-7)
    ti = ti.cpu().numpy()
    bad = np.nonzero((ti != ri).any(axis=1))[0]
    for r in bad:   # only f32 near-ties (<= 4 ulp) may swap
        assert np.all(np.abs(s64[r, ti[r]] - rv[r]) <= 4 * np.spacing(np.float32(1.0)))
    assert len(bad) <= 1


def test_c2_range_db_med_10k_queries(tmp_path):
    N, B = synth.BANK_ROWS["range_db_med"], 10_000
    m, obank, w = _model(tmp_path, N)
    q = synth.make_queries(B, seed=7)
    x = torch.from_numpy(q).to("cuda:0")
    out = m(x, return_device=True)
    assert m.engine.kept_queries() == B                  # pass 2 ran on the kept logits
    qt, ns = m.engine.last_geometry()
    assert qt == 157 and ns > 1                          # the split geometry of a 10k-query batch
    _check_all_rows(out, m, x, float(obank.values.min()), float(obank.values.max()))
    _check_sample(out, m, q, x, obank, w)
    # the numpy contract (range_forward_host: pass 2 in two parts, host array filled slab by slab
    # while later slabs are still being computed / copied) returns the same values: within
    # split-order rounding for large batches, bit-identical for one-part batches
    host = m(x)
    assert isinstance(host, np.ndarray) and host.dtype == np.float64 and host.shape == (B, 1280)
    np.testing.assert_allclose(host, out.cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert np.array_equal(host[:, 1024:], out[:, 1024:].cpu().numpy())
    small = m(x[:3000])
    assert np.array_equal(small, m(x[:3000], return_device=True).cpu().numpy())
    np.testing.assert_allclose(small, host[:3000], rtol=1e-5, atol=1e-5)


def test_c3_range_db_large_100k_queries(tmp_path):
    N, B = synth.BANK_ROWS["range_db_large"], 100_000
    m, obank, w = _model(tmp_path, N)
    # queries over the whole sphere (the e-hat of high latitudes is the exact-math value, see
    # DESIGN.md section 4; the retrieval half is compared given the oracle's own e-hat)
    q = synth.make_queries(B, seed=7, lat_max=90.0)
    x = torch.from_numpy(q).to("cuda:0")
    out = m(x, return_device=True)
    assert m.chunk_size == 16384 and m.engine.kept_queries() == B - 6 * 16384
    _check_all_rows(out, m, x, float(obank.values.min()), float(obank.values.max()))
    _check_sample(out, m, q, x, obank, w)
    # one 10 000-query batch = the benchmark's launch (157 query tiles x 13 bank splits)
    out10k = m(x[:10_000], return_device=True)
    assert m.engine.last_geometry() == (157, 13) and m.engine.kept_queries() == 10_000
    # (other split boundaries: float32 sums of ~10^4 terms in another order, worst element of 1.3e7 at ~40 ulp)
    np.testing.assert_allclose(out10k.cpu().numpy(), out[:10_000].cpu().numpy(), rtol=1e-5, atol=1e-5)
    # the same launch again: bit-identical (fixed summation orders everywhere; a race would show)
    for _ in range(3):
        assert torch.equal(m(x[:10_000], return_device=True), out10k)
    # the opt-in arithmetic of pass 2 (pv_mode='bf16x3') at the same launch, EVERY output against
    # the exact kernel, several launches (a race between the waves of a workgroup showed in two or
    # three of the 2 041 workgroups per launch, never in small cases)
    m.engine.set_pv_mode("bf16x3")
    try:
        first = None
        for _ in range(4):
            fast = m(x[:10_000], return_device=True)
            first = fast if first is None else first
            assert torch.equal(fast, first)
            assert m.engine.last_geometry() == (157, 13)
            assert torch.equal(fast[:, 1024:], out10k[:, 1024:])
            d = (fast[:, :1024] - out10k[:, :1024]).abs()
            # (the two planted constant columns are float32 sums of 10^5 same-sign terms: another
            # order of summation shows at ~15 ulp there, 4e-6 on -2.5; elsewhere 2e-7)
            assert float(d[:, 2:].max()) < 1e-6 and float(d[:, :2].max()) < 1e-5
        _check_all_rows(fast, m, x[:10_000], float(obank.values.min()), float(obank.values.max()))
    finally:
        m.engine.set_pv_mode("exact")


def test_small_batches_at_full_size(tmp_path):
    """The latency path at the benchmark's sizes (range_db_large, L = 40, H = 512): 1 / 16 / 17 / 32
    queries run the persistent encoder launch and ONE pass over the bank (attend_small.h, one or two
    query tiles per workgroup), 33 / 64 / 128 / 512 queries the persistent encoder and the two-pass kernels,
    1 250 (a rank of 8's share of BASELINE's batch) the persistent encoder at 79 tiles (its parts taken in turns).
    Every row against the float64 oracle and the reference's float32 order, against the same rows of
    one large batch (other kernels, other summation orders: float32 rounding), bit-identical when
    repeated, and the numpy contract returns the same bits."""
    N = synth.BANK_ROWS["range_db_large"]
    m, obank, w = _model(tmp_path, N)
    q = synth.make_queries(2304, seed=11)
    x = torch.from_numpy(q).to("cuda:0")
    big = m(x, return_device=True).cpu().numpy()
    for B in (1, 16, 17, 32, 33, 64, 128, 512, 1250):
        m.engine.profile_enable(True)
        out = m(x[:B], return_device=True)
        one_pass = m.engine.profile_read(1)[1] == 0          # no pass 1 ran
        m.engine.profile_enable(False)
        assert one_pass == (B <= 32)
        for _ in range(3):
            assert torch.equal(m(x[:B], return_device=True), out)
        got = out.cpu().numpy()
        e = got[:, 1024:]
        np.testing.assert_allclose(e, big[:B, 1024:], rtol=0, atol=1e-13)
        np.testing.assert_allclose(got[:, :1024], O.retrieve64(e, q[:B], obank, "RANGE+", 0.5), rtol=0, atol=2e-5)
        np.testing.assert_allclose(got, O.retrieve(e, q[:B], obank, "RANGE+", 0.5), rtol=0, atol=1e-4)
        np.testing.assert_allclose(got[:, :1024], big[:B, :1024], rtol=0, atol=1e-5)
        assert np.abs(got[:, 0] - 1.0).max() < 1e-5 and np.abs(got[:, 1] + 2.5).max() < 2.5e-5
        assert np.array_equal(m(x[:B]), got)


def test_bench_self_launches_two_ranks():
    """``python bench.py --gpus 2`` from a plain shell starts its own rank processes (fresh
    children, torch.distributed.run) and relays rank 0's line.  Two gloo ranks share the one GPU
    of the test box (RCCL needs a GPU per rank): the timing means nothing, the flow is the N>1 one."""
    env = dict(os.environ, RANGE_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--bank", "range_db_med"],
                       env=env, cwd=REPO, capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "strong"
    assert r["config"]["queries_total"] == 10_000 and r["config"]["queries_per_gpu"] == 5_000
    assert r["dist"]["backend"] == "gloo" and r["dist"]["world_size"] == 2
    assert r["parity_max_abs"] < 1e-4
    assert r["weak"]["queries_per_gpu"] == 10_000
    assert r["control_query_sharded"]["max_abs_vs_row_sharded"] < 1e-5     # two layouts, one answer
    # a rank that fails makes the launcher exit non-zero
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--queries", "10001"],
                       env=env, cwd=REPO, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def test_c5_beta_sweep_full_bank(tmp_path):
    """BASELINE config 5 on one GPU: the beta sweep against range_db_large, 20 000 queries (two
    chunks), every beta against the float64 oracle on a sample, every row through the planted
    columns, and the beta = 0.5 slice against the plain forward."""
    N, B = synth.BANK_ROWS["range_db_large"], 20_000
    betas = (0.0, 0.25, 0.5, 0.75, 1.0)
    m, obank, w = _model(tmp_path, N)
    q = synth.make_queries(B, seed=11)
    x = torch.from_numpy(q).to("cuda:0")
    sw = m.sweep(x, betas, return_device=True)
    assert sw.shape == (len(betas), B, 1280) and sw.dtype == torch.float64 and sw.is_cuda
    vmin, vmax = float(obank.values.min()), float(obank.values.max())
    idx = np.sort(np.random.default_rng(2).choice(B, 48, replace=False))
    sel = torch.from_numpy(idx).to("cuda:0")
    for j, b in enumerate(betas):
        # (the geographic weights are spread over far more rows than the semantic ones: more float32
        # terms of similar size per sum, mean error of the planted column 3.5e-6 at beta = 0)
        _check_all_rows(sw[j], m, x, vmin, vmax, mean_tol=6e-6)
        got = sw[j][sel].cpu().numpy()
        ref = O.retrieve64(got[:, 1024:], q[idx], obank, "RANGE+", b)
        np.testing.assert_allclose(got[:, :1024], ref, rtol=0, atol=2e-5)
    # the blend of H and G (range.py:238, float32) against the forward's one combined weight
    fwd = m(x, return_device=True)
    assert torch.equal(sw[2][:, 1024:], fwd[:, 1024:])
    d = (sw[2][:, :1024] - fwd[:, :1024]).abs()
    # (planted constant columns: float32 sums of 10^5 same-sign terms, formed once per retrieval
    # here and once for the combined weight there - 2.3e-5 on -2.5; other columns 1e-7)
    assert float(d[:, 2:].max()) < 2e-6 and float(d[:, :2].max()) < 5e-5
    # beta = 1 is the semantic retrieval alone, beta = 0 the geographic one: they differ
    assert float((sw[0] - sw[4]).abs().max()) > 1e-3


def test_c5_beta_sweep_one_million_queries(tmp_path):
    """BASELINE config 5 at its FULL query count on one GPU: 1 000 000 queries x 5 betas over
    range_db_large (51 GB of float64 results, held in HBM: 288 GB per GPU is what makes that a
    non-event).  Every one of the 5 x 10^6 rows through the size-independent properties (planted
    constant columns reproduced iff a row's weights sum to one, value range, unit e-hat, finite),
    a sample of every beta against the float64 oracle, and the sweep of a slab again, bit for bit."""
    N, B = synth.BANK_ROWS["range_db_large"], 1_000_000
    betas = (0.0, 0.25, 0.5, 0.75, 1.0)
    m, obank, w = _model(tmp_path, N)
    q = synth.make_queries(B, seed=77)
    x = torch.from_numpy(q).to("cuda:0")
    sw = m.sweep(x, betas, return_device=True)
    assert sw.shape == (len(betas), B, 1280) and sw.dtype == torch.float64 and sw.is_cuda
    vmin, vmax = float(obank.values.min()), float(obank.values.max())
    for j, b in enumerate(betas):
        worst0 = worst1 = 0.0
        for lo in range(0, B, 100_000):           # (slabs: the reductions' temporaries stay small)
            s = sw[j, lo:lo + 100_000]
            assert bool(torch.isfinite(s).all())
            worst0 = max(worst0, float((s[:, 0] - 1.0).abs().max()))
            worst1 = max(worst1, float((s[:, 1] + 2.5).abs().max()))
            assert float(s[:, :1024].max()) <= vmax and float(s[:, :1024].min()) >= vmin
            assert float((s[:, 1024:].norm(dim=1) - 1.0).abs().max()) < 1e-12
        assert worst0 < 1e-5 and worst1 < 2.5e-5, (b, worst0, worst1)
        assert torch.equal(sw[j, :, 1024:], sw[0, :, 1024:])      # e-hat does not depend on beta
    idx = np.sort(np.random.default_rng(3).choice(B, 32, replace=False))
    sel = torch.from_numpy(idx).to("cuda:0")
    for j, b in enumerate(betas):
        got = sw[j][sel].cpu().numpy()
        np.testing.assert_allclose(got[:, :1024], O.retrieve64(got[:, 1024:], q[idx], obank, "RANGE+", b), rtol=0, atol=2e-5)
    # a slab of the job on its own: the same bits (chunks of 16 384 queries: the slab starts on a chunk)
    lo = 16384 * 7
    again = m.sweep(x[lo:lo + 16384], betas, return_device=True)
    assert torch.equal(again, sw[:, lo:lo + 16384])


@pytest.mark.parametrize("keep", [True, False])
def test_dram_sized_bank_through_load_model(keep, tmp_path, monkeypatch):
    """N = 10^6 rows (5.1 GB of bank: ten times range_db_large, the values far beyond the 256 MB
    Infinity Cache) through ``load_model(...)`` on BOTH forms of pass 2 - on the logits pass 1 kept
    (16 GB for the 4 096 queries of a call) and recomputing them (RANGE_KEEP_LOGITS=0: what a call
    does whose logits no longer fit in half of the free memory): every row through the
    planted-column properties, a sample against the float64 oracle over the whole bank.  (The bank
    is generated on the device and handed to ``load_model`` in place of the file reader: no 5 GB file.)"""
    import range_amd.range as R
    from range_amd.bank import PreparedBank
    N, B = 1_000_000, 4096
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    keys = torch.nn.functional.normalize(torch.randn((N, 256), generator=g, device=dev), dim=1)
    vals = torch.randn((N, 1024), generator=g, device=dev)
    vals[:, 0] = 1.0
    vals[:, 1] = -2.5
    lon = torch.rand(N, generator=g, device=dev, dtype=torch.float64) * 360 - 180
    lat = torch.asin(torch.rand(N, generator=g, device=dev, dtype=torch.float64) * 2 - 1) * 180 / np.pi
    rad = torch.stack([lon, lat], 1).float() * np.float32(np.pi / 180)
    xyz = torch.stack([torch.cos(rad[:, 1]) * torch.cos(rad[:, 0]), torch.cos(rad[:, 1]) * torch.sin(rad[:, 0]), torch.sin(rad[:, 1])], 1)
    bank = PreparedBank(keys.cpu().numpy(), vals.cpu().numpy(), xyz.float().cpu().numpy())
    del keys, vals, xyz, rad
    torch.cuda.empty_cache()
    monkeypatch.setattr(R, "load_bank", lambda path: bank)
    if not keep:
        monkeypatch.setenv("RANGE_KEEP_LOGITS", "0")
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=SEED)
    m = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path="in-memory", beta=0.5)
    q = synth.make_queries(B, seed=3, lat_max=90.0)
    out = m(torch.from_numpy(q).to(dev), return_device=True)
    assert out.shape == (B, 1280) and bool(torch.isfinite(out).all())
    assert (m.engine.kept_queries() == B) == keep
    # float32 sums over 10^6 weights: the worst of 4 096 rows (tools/big_bank.py)
    assert float((out[:, 0] - 1.0).abs().max()) < 5e-5 and float((out[:, 1] + 2.5).abs().max()) < 1e-4
    assert float((out[:, 1024:].norm(dim=1) - 1.0).abs().max()) < 1e-12
    idx = np.linspace(0, B - 1, 12).astype(np.int64)
    got = out[torch.from_numpy(idx).to(dev)].cpu().numpy()
    obank = O.Bank(bank.keys, bank.values, bank.xyz)
    ref64 = O.retrieve64(got[:, 1024:], q[idx], obank, "RANGE+", 0.5)
    assert float(np.abs(got[:, :1024] - ref64).max()) < 2e-5
