"""GPU parity tests of the ridge probe (include/range_probe.h) against the CPU oracle
(oracle/probe_oracle.py) and the values returned by the reference's own evaluate_npz
(tests/golden/probe_*.npz).  Run on an MI355X with ``pytest -m gpu``."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import probe_oracle as po
from range_amd import evaluate as ev
from tools import synth
from range_amd._probe_native import ProbeEngine

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dev(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to("cuda:0").contiguous()


@pytest.mark.parametrize("M,N,K,ta,tb", [
    (1, 1, 1, False, False), (16, 16, 4, False, False), (130, 70, 33, False, False),
    (257, 129, 100, True, False), (64, 200, 515, False, True), (300, 300, 65, True, True),
    (128, 128, 5000, True, False),          # split-K slabs
    (1280, 3, 64, False, False), (5, 1280, 64, True, False),
])
def test_gemm_vs_numpy(M, N, K, ta, tb):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    C0 = rng.standard_normal((M, N))
    eng = ProbeEngine("cuda:0")
    ref = (A.T if ta else A) @ (B.T if tb else B)
    tol = 1e-13 * K ** 0.5 * 8
    got = eng.gemm(_dev(A), _dev(B), ta, tb).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=tol)
    out = _dev(C0)
    eng.gemm(_dev(A), _dev(B), ta, tb, alpha=-0.5, beta=2.0, out=out)
    np.testing.assert_allclose(out.cpu().numpy(), -0.5 * ref + 2.0 * C0, rtol=0, atol=2 * tol)


@pytest.mark.parametrize("n,rows", [(64, 100), (200, 4000), (1280, 3000)])
def test_syrk_lower_only(n, rows):
    rng = np.random.default_rng(n + rows)
    Z = rng.standard_normal((rows, n))
    eng = ProbeEngine("cuda:0")
    out = torch.full((n, n), float("nan"), dtype=torch.float64, device="cuda:0")
    eng.gemm(_dev(Z), _dev(Z), True, False, out=out, lower_only=True)
    got = out.cpu().numpy()
    ref = Z.T @ Z
    il = np.tril_indices(n)
    np.testing.assert_allclose(got[il], ref[il], rtol=0, atol=1e-13 * rows ** 0.5 * 8)


def test_column_kernels_vs_numpy():
    rng = np.random.default_rng(1)
    eng = ProbeEngine("cuda:0")
    for n, d in ((1, 1), (77, 130), (5000, 1280), (100000, 3)):
        X = rng.standard_normal((n, d)) * 3 + 1
        mn, mx, sm = (t.cpu().numpy() for t in eng.colstats(_dev(X)))
        np.testing.assert_array_equal(mn, X.min(axis=0))
        np.testing.assert_array_equal(mx, X.max(axis=0))
        np.testing.assert_allclose(sm, X.sum(axis=0), rtol=1e-12, atol=1e-9)
        scale, off = po.minmax_fit(X)
        perm = rng.permutation(n)[: max(1, n // 2)]
        shift = rng.standard_normal(d)
        got = eng.scale_rows(_dev(X), _dev(perm, torch.int64), _dev(scale), _dev(off),
                             _dev(shift)).cpu().numpy()
        np.testing.assert_array_equal(got, po.minmax_apply(X[perm], scale, off) - shift)
        np.testing.assert_array_equal(eng.scale_rows(_dev(X)).cpu().numpy(), X)
    code = rng.integers(0, 7, size=1000).astype(np.int32)
    shift = rng.standard_normal(7)
    T = eng.onehot(_dev(code, torch.int32), 7, 0, _dev(shift)).cpu().numpy()
    np.testing.assert_array_equal(T, np.where(code[:, None] == np.arange(7)[None], 1.0, -1.0) - shift)
    T1 = eng.onehot(_dev(code, torch.int32), 1, 1, _dev(shift[:1])).cpu().numpy()
    np.testing.assert_array_equal(T1[:, 0], np.where(code == 1, 1.0, -1.0) - shift[0])


@pytest.mark.parametrize("n,d,c,k", [(300, 40, 2, 3), (900, 200, 5, 3), (2000, 1280, 1, 3),
                                     (700, 96, 70, 10), (150, 65, 3, 1)])
def test_batched_ridge_solve_vs_oracle(n, d, c, k):
    """Every (fold, alpha) coefficient vector and intercept against oracle.ridge_fit on the same
    training rows (k == 1: the refit on all rows)."""
    rng = np.random.default_rng(n + d)
    X = rng.uniform(0, 1, size=(n, d)) * rng.uniform(0.2, 1.0, size=d)
    Y = X @ rng.standard_normal((d, c)) + 0.1 * rng.standard_normal((n, c)) + 3.0
    eng = ProbeEngine("cuda:0")
    shift_x, shift_y = X.mean(axis=0) + 0.01, Y.mean(axis=0) - 0.02     # any origin must work
    Z, T = _dev(X - shift_x), _dev(Y - shift_y)
    folds = po.kfold_ids(n, k) if k > 1 else np.zeros(n, dtype=np.int64)
    edges = np.concatenate([[0], np.cumsum(np.bincount(folds))])
    kk = max(k, 1)
    Gf, Bf = eng.empty((kk, d, d)), eng.empty((kk, d, c))
    zs, ts = eng.empty((kk, d)), eng.empty((kk, c))
    for f in range(kk):
        eng.gram(Z[edges[f]:edges[f + 1]], T[edges[f]:edges[f + 1]], Gf[f], Bf[f], zs[f], ts[f])
    Gt, Bt, zt, tt = eng.sum_parts(Gf), eng.sum_parts(Bf), eng.sum_parts(zs), eng.sum_parts(ts)
    if k > 1:
        ntr = [float(n - (edges[f + 1] - edges[f])) for f in range(k)]
        W, c0 = eng.solve(Gt, Bt, zt, tt, ntr, po.ALPHAS, Gf, Bf, zs, ts)
    else:
        W, c0 = eng.solve(Gt, Bt, zt, tt, [float(n)], po.ALPHAS)
    W, c0 = W.cpu().numpy(), c0.cpu().numpy()
    for f in range(kk):
        tr = folds != f if k > 1 else np.ones(n, bool)
        for a, alpha in enumerate(po.ALPHAS):
            Wr, br = po.ridge_fit(X[tr], Y[tr], alpha)
            np.testing.assert_allclose(W[f, :, a, :], Wr, rtol=1e-7, atol=1e-9)
            # intercept in the shifted coordinates: y - shift_y = (x - shift_x).W + c0
            np.testing.assert_allclose(c0[f, a], br - shift_y + shift_x @ Wr, rtol=1e-7, atol=1e-9)


def test_solve_reports_indefinite_system():
    eng = ProbeEngine("cuda:0")
    d, c = 70, 1
    G = -np.eye(d)
    with pytest.raises(RuntimeError, match="positive definite"):
        eng.solve(_dev(G), _dev(np.ones((d, c))), _dev(np.zeros(d)), _dev(np.zeros(c)), [10.0],
                  [0.5])


@pytest.mark.parametrize("tag", sorted(synth.PROBE_CASES))
def test_probe_vs_reference_golden_and_oracle(tag):
    task_name, kw = synth.PROBE_CASES[tag]
    g = np.load(os.path.join(GOLDEN, f"probe_{tag}.npz"))
    t = synth.make_probe_task(**kw)
    classification = kw["kind"] == "classification"
    r = ev.RidgeProbe("cuda:0").fit_score(t["train_embeddings"], t["train_y"], t["val_embeddings"],
                                          t["val_y"], classification)
    o = po.probe(t["train_embeddings"], t["train_y"], t["val_embeddings"], t["val_y"],
                 po.task_kind(task_name))
    assert r["alpha"] == float(g["alpha"]) == o["alpha"]
    if classification:
        # accuracies are ratios of counts: identical unless an arg-max is tied to rounding
        assert r["score"] == float(g["score"])
        np.testing.assert_array_equal(r["cv_scores"], o["cv_scores"])
    else:
        # float64 throughout; the normal equations (scikit-learn's own route when rows >=
        # features) lose about cond(A)*eps: 1e-9 on R^2 is the stated tolerance
        assert abs(r["score"] - float(g["score"])) < 1e-9
        np.testing.assert_allclose(r["cv_scores"], o["cv_scores"], rtol=0, atol=1e-9)


def test_evaluate_npz_drop_in(tmp_path, capsys):
    task_name, kw = synth.PROBE_CASES["cls_checker"]
    g = np.load(os.path.join(GOLDEN, "probe_cls_checker.npz"))
    synth.write_probe_task(str(tmp_path), "RANGE+", task_name, synth.make_probe_task(**kw))
    args = Namespace(embeddings_dir=str(tmp_path), location_model_name="RANGE+",
                     task_name=task_name, device="cuda:0")
    acc = ev.evaluate_npz(args)
    out = capsys.readouterr().out
    assert acc == float(g["score"])
    assert out.splitlines()[0] == str(g["banner"]) == "Classification Model"
    assert f"The validation set accuracy is {acc:3f}" in out
    task_name, kw = synth.PROBE_CASES["reg_d64"]
    g = np.load(os.path.join(GOLDEN, "probe_reg_d64.npz"))
    synth.write_probe_task(str(tmp_path), "RANGE+", task_name, synth.make_probe_task(**kw))
    args = Namespace(embeddings_dir=str(tmp_path), location_model_name="RANGE+", task_name=task_name)
    assert abs(ev.evaluate_npz(args) - float(g["score"])) < 1e-9
    assert capsys.readouterr().out.splitlines()[0] == "Regression Model"
    with pytest.raises(AssertionError):
        ev.evaluate_npz(Namespace(embeddings_dir=str(tmp_path), location_model_name="RANGE+",
                                  task_name="absent"))


def test_pipeline_embed_save_probe(tmp_path, capsys):
    """The reference's two-step workflow end to end on the GPU (range/range.py:281-306):
    save_embeddings with a RANGE+ model, then evaluate_npz on the files it wrote; the probe's
    score equals the oracle's on the same files; the plain encoders go through the same driver."""
    from range_amd import load_model
    from range_amd.save import save_embeddings
    ck = synth.write_checkpoint(str(tmp_path / "enc.ckpt"), L=10, hidden=64, seed=5)
    db = synth.write_bank(str(tmp_path / "db.npz"), 2000, 3)
    rng = np.random.default_rng(12)
    q = synth.make_queries(2600, seed=41, lat_max=80.0)
    # synthetic downstream targets that depend smoothly on the location
    lat, lon = np.deg2rad(q[:, 1]), np.deg2rad(q[:, 0])
    temp = 25.0 * np.cos(lat) + 3.0 * np.sin(2 * lon) + rng.normal(0, 0.5, q.shape[0])
    biome = (np.digitize(q[:, 1], [-40, -10, 10, 40]) * 2 + (q[:, 0] > 0)).astype(np.int64)

    def loader(y, a, b, bs=512):
        return [(torch.from_numpy(q[i:min(i + bs, b)]), torch.from_numpy(y[i:min(i + bs, b)]))
                for i in range(a, b, bs)]

    emb_dir = str(tmp_path / "emb")
    for name, kw in (("RANGE+", dict(db_path=db, beta=0.5)), ("Wrap", {})):
        model = load_model(name, pretrained_path=ck, device="cuda:0", **kw)
        for task, y in (("temperature", temp), ("biome", biome)):
            args = Namespace(embeddings_dir=emb_dir, location_model_name=name, task_name=task,
                             device="cuda:0")
            save_embeddings(args, loader(y, 0, 2000), loader(y, 2000, 2600), model)
            score = ev.evaluate_npz(args)
            tr = np.load(os.path.join(emb_dir, name, task + "_train.npz"))
            va = np.load(os.path.join(emb_dir, name, task + "_val.npz"))
            assert tr["embeddings"].shape == (2000, model.location_feature_dim)
            ref = po.probe(tr["embeddings"], tr["y"], va["embeddings"], va["y"], po.task_kind(task))
            if task == "biome":
                assert score == ref["score"] and score > 0.5
            else:
                assert abs(score - ref["score"]) < 1e-9 and score > 0.5
    capsys.readouterr()
