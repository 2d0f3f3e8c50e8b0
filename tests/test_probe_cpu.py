"""CPU checks of the rows after the hot path: the probe oracle against the reference's own
``evaluate_npz`` results (tests/golden/probe_*.npz, written by make_golden_next.py) and against
scikit-learn, and the coordinate-encoder oracle against the reference's outputs."""
import os
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import probe_oracle as po  # noqa: E402
from oracle import range_oracle as ro  # noqa: E402
from tools import synth  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def test_coord_oracle_matches_reference():
    g = np.load(os.path.join(GOLD, "coord_encoders.npz"))
    for name, dim in (("Direct", 2), ("Cartesian_3D", 3), ("Wrap", 4)):
        got = ro.coord_features(g["lonlat"], name)
        assert got.shape == (g["lonlat"].shape[0], dim) and int(g[name + "_dim"]) == dim
        # numpy and torch CPU share libm/Sleef to the last ulp or so
        np.testing.assert_allclose(got, g[name], rtol=0, atol=3e-16)
    assert str(g["Direct_type"]) == "Tensor" and str(g["Wrap_type"]) == "Tensor"
    assert str(g["Cartesian_3D_type"]) == "ndarray"
    np.testing.assert_array_equal(ro.coord_features(g["lonlat"], "Direct"), g["Direct"])


def test_task_dispatch():
    for t in ("ecoregion", "biome", "country", "ocean", "checker_5", "my_checker"):
        assert po.task_kind(t) == "classification"
    for t in ("elevation", "population", "temperature"):
        assert po.task_kind(t) == "regression"
    with pytest.raises(NotImplementedError):
        po.task_kind("inat_2018")


@pytest.mark.parametrize("tag", sorted(synth.PROBE_CASES))
def test_probe_oracle_matches_reference(tag):
    task_name, kw = synth.PROBE_CASES[tag]
    g = np.load(os.path.join(GOLD, f"probe_{tag}.npz"))
    assert str(g["task_name"]) == task_name
    t = synth.make_probe_task(**kw)
    r = po.probe(t["train_embeddings"], t["train_y"], t["val_embeddings"], t["val_y"],
                 po.task_kind(task_name))
    assert r["alpha"] == float(g["alpha"])
    if kw["kind"] == "classification":
        assert r["score"] == float(g["score"])              # a ratio of counts
    else:
        assert abs(r["score"] - float(g["score"])) < 1e-10
    assert abs(r["cv_scores"].mean(axis=1).max() - float(g["best_cv_score"])) < 1e-10


def test_folds_match_sklearn():
    from sklearn.model_selection import KFold, StratifiedKFold
    rng = np.random.default_rng(3)
    for n, k in ((10, 3), (11, 3), (12, 3), (1000, 3), (7, 7)):
        ids = po.kfold_ids(n, k)
        for f, (_, te) in enumerate(KFold(k).split(np.zeros(n))):
            np.testing.assert_array_equal(np.nonzero(ids == f)[0], te)
    for n, c, k in ((200, 3, 10), (503, 17, 10), (64, 2, 10), (300, 40, 10), (90, 5, 3)):
        y = rng.choice(np.arange(c) * 5 - 3, size=n, p=rng.dirichlet(np.ones(c)))
        ids = po.stratified_kfold_ids(y, k)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for f, (_, te) in enumerate(StratifiedKFold(k).split(np.zeros(n), y)):
                np.testing.assert_array_equal(np.nonzero(ids == f)[0], te)


def test_ridge_and_scaler_match_sklearn():
    from sklearn.linear_model import Ridge, RidgeClassifier
    from sklearn.preprocessing import MinMaxScaler
    rng = np.random.default_rng(4)
    X = rng.standard_normal((300, 40)) * 3 + 1
    X[:, 5] = 2.5                                              # a constant column
    sc = MinMaxScaler().fit(X)
    scale, off = po.minmax_fit(X)
    np.testing.assert_array_equal(po.minmax_apply(X, scale, off), sc.transform(X))
    Xs = sc.transform(X)
    Y = rng.standard_normal((300, 2))
    for n in (300, 30):                                        # primal and dual branch
        for alpha in po.ALPHAS:
            W, b = po.ridge_fit(Xs[:n], Y[:n], alpha)
            m = Ridge(alpha=alpha).fit(Xs[:n], Y[:n])
            np.testing.assert_allclose(W.T, m.coef_, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(b, m.intercept_, rtol=1e-9, atol=1e-12)
    y = rng.integers(0, 4, size=300) * 2 + 1
    for labels in (y, (y > 3).astype(np.int64)):
        pred = po.classifier_fit_predict(Xs[:200], labels[:200], Xs[200:], 1.0)
        ref = RidgeClassifier(alpha=1.0).fit(Xs[:200], labels[:200]).predict(Xs[200:])
        np.testing.assert_array_equal(pred, ref)


def test_product_fold_assignment_matches_sklearn():
    """The host-side fold logic of range_amd/evaluate.py (indices only, no arithmetic)."""
    from sklearn.model_selection import KFold, StratifiedKFold
    from range_amd import evaluate as ev
    rng = np.random.default_rng(5)
    for n, k in ((10, 3), (11, 3), (12, 3), (1000, 3), (7, 7), (29, 10)):
        ids = ev.kfold_ids(n, k)
        np.testing.assert_array_equal(ids, po.kfold_ids(n, k))
        for f, (_, te) in enumerate(KFold(k).split(np.zeros(n))):
            np.testing.assert_array_equal(np.nonzero(ids == f)[0], te)
    for n, c, k in ((200, 3, 10), (503, 17, 10), (64, 2, 10), (300, 40, 10), (90, 5, 3)):
        y = rng.choice(np.arange(c) * 5 - 3, size=n, p=rng.dirichlet(np.ones(c)))
        ids = ev.stratified_kfold_ids(y, k)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for f, (_, te) in enumerate(StratifiedKFold(k).split(np.zeros(n), y)):
                np.testing.assert_array_equal(np.nonzero(ids == f)[0], te)
    with pytest.raises(ValueError):
        ev.stratified_kfold_ids(np.array([0, 1, 2, 0, 1, 2]), 3 + 1)
    assert ev._is_classification("checker_9") and not ev._is_classification("elevation")


def test_probe_requires_gpu(tmp_path):
    import torch
    from argparse import Namespace
    from range_amd import evaluate as ev
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    task_name, kw = synth.PROBE_CASES["reg_d64"]
    synth.write_probe_task(str(tmp_path), "RANGE+", task_name, synth.make_probe_task(**kw))
    args = Namespace(embeddings_dir=str(tmp_path), location_model_name="RANGE+", task_name=task_name)
    with pytest.raises(RuntimeError):              # no CPU fallback: the loader fails loudly
        ev.evaluate_npz(args)
    with pytest.raises(AssertionError):            # evaluate.py:17-18
        ev.evaluate_npz(Namespace(embeddings_dir=str(tmp_path), location_model_name="RANGE+",
                                  task_name="missing"))
    with pytest.raises(NotImplementedError):       # evaluate.py:31-32
        synth.write_probe_task(str(tmp_path), "RANGE+", "inat_2018", synth.make_probe_task(**kw))
        ev.evaluate_npz(Namespace(embeddings_dir=str(tmp_path), location_model_name="RANGE+",
                                  task_name="inat_2018"))
