"""CPU oracle for the downstream ridge probe - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench/tools CPU-baseline legs may import this module; the
product path (range_amd/evaluate.py -> librange_hip.so) never does.

What it restates
----------------
``evaluate_npz`` of the reference (range/utils/evaluate.py:14-47):

    scaler = MinMaxScaler(); X = scaler.fit_transform(train); Xv = scaler.transform(val)   (:38-42)
    regression tasks    : RidgeCV(alphas=(0.1, 1.0, 10.0), cv=3)                            (:35)
    classification tasks: RidgeClassifierCV(alphas=(0.1, 1.0, 10.0), cv=10)                 (:30)
    clf.fit(X, y); return clf.score(Xv, yv)                                                 (:44-45)

The arithmetic lives in a third-party dependency, scikit-learn (pinned ``scikit-learn==1.2.0`` in
the reference's requirements.txt:15; 1.7.2 is what this image has - the code path below is the
same in both).  With an integer ``cv`` both estimators run ``GridSearchCV`` over ``alpha`` with a
plain ``Ridge`` / ``RidgeClassifier`` per (fold, alpha), pick the alpha with the best mean
held-out score (first one on ties), and refit on all rows:

* folds: ``KFold(cv)`` without shuffling for regression, ``StratifiedKFold(cv)`` without shuffling
  for classification (sklearn.model_selection.check_cv);
* ``Ridge(fit_intercept=True, solver='auto')`` on dense input = centre X and y by the training
  means, solve ``(XcT Xc + alpha I) w = XcT yc`` by Cholesky (features <= rows) or the dual
  ``(Xc XcT + alpha I) a = yc, w = XcT a`` (features > rows); intercept = ybar - xbar.w;
* ``RidgeClassifier``: labels -> one column of -1/+1 per class of the TRAINING rows (one column
  when two classes), the same ridge per column, predict argmax (or ``score > 0``);
* scores: R^2 (uniform average over target columns) / accuracy.

Pinned by tests/golden/probe_*.npz: values returned by the reference's own ``evaluate_npz`` run in
this container on the seeded tasks of tools.synth.PROBE_CASES (tests/golden/make_golden_next.py)
and, in tests/test_probe_cpu.py, by scikit-learn itself.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import scipy.linalg

ALPHAS = (0.1, 1.0, 10.0)                      # evaluate.py:30, :35
CLASSIFICATION_TASKS = ("ecoregion", "biome", "country", "ocean")   # evaluate.py:28


def task_kind(task_name: str) -> str:
    """The dispatch of evaluate.py:28-35."""
    if task_name in CLASSIFICATION_TASKS or "checker" in task_name:
        return "classification"
    if "inat" in task_name:
        raise NotImplementedError("Inat evaluation not implemented")
    return "regression"


# ---- MinMaxScaler (sklearn.preprocessing.MinMaxScaler, feature_range (0,1)) --------------------
def minmax_fit(X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    lo = X.min(axis=0)
    rng = X.max(axis=0) - lo
    rng = np.where(rng < 10 * np.finfo(rng.dtype).eps, 1.0, rng)    # constant columns
    scale = 1.0 / rng
    return scale, 0.0 - lo * scale


def minmax_apply(X: np.ndarray, scale: np.ndarray, offset: np.ndarray) -> np.ndarray:
    out = X * scale
    out += offset
    return out


# ---- folds ---------------------------------------------------------------------------------------
def kfold_ids(n: int, k: int) -> np.ndarray:
    """Fold id per row for KFold(k, shuffle=False): contiguous blocks, the first n % k one longer."""
    sizes = np.full(k, n // k, dtype=np.int64)
    sizes[: n % k] += 1
    return np.repeat(np.arange(k), sizes)


def stratified_kfold_ids(y: np.ndarray, k: int) -> np.ndarray:
    """Fold id per row for StratifiedKFold(k, shuffle=False).

    Classes are numbered by first appearance; the sorted label sequence is dealt round-robin to the
    folds, which fixes how many rows of each class every fold receives; each class then hands its
    rows, in order of appearance, to fold 0 first, then fold 1, ..."""
    _, first, inv = np.unique(y, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first))             # class -> rank of first appearance
    code = order[inv]
    n_classes = first.size
    counts = np.bincount(code, minlength=n_classes)
    if np.all(k > counts):
        raise ValueError("n_splits cannot be greater than the number of members in each class.")
    dealt = np.sort(code)
    quota = np.stack([np.bincount(dealt[i::k], minlength=n_classes) for i in range(k)])
    folds = np.empty(y.shape[0], dtype=np.int64)
    for c in range(n_classes):
        folds[code == c] = np.repeat(np.arange(k), quota[:, c])
    return folds


# ---- ridge ---------------------------------------------------------------------------------------
def ridge_fit(X: np.ndarray, Y: np.ndarray, alpha: float) -> Tuple[np.ndarray, np.ndarray]:
    """Ridge(alpha, fit_intercept=True) on dense X (n,d), Y (n,c) -> (W (d,c), intercept (c,))."""
    xbar = X.mean(axis=0)
    ybar = Y.mean(axis=0)
    Xc = X - xbar
    Yc = Y - ybar
    n, d = Xc.shape
    if d > n:
        K = Xc @ Xc.T
        K[np.diag_indices_from(K)] += alpha
        W = Xc.T @ scipy.linalg.solve(K, Yc, assume_a="pos")
    else:
        A = Xc.T @ Xc
        A[np.diag_indices_from(A)] += alpha
        W = scipy.linalg.solve(A, Xc.T @ Yc, assume_a="pos")
    return W, ybar - xbar @ W


def binarize(y: np.ndarray, classes: np.ndarray) -> np.ndarray:
    """LabelBinarizer(pos_label=1, neg_label=-1): (n, C) of -1/+1, or (n, 1) when C == 2."""
    Y = np.where(y[:, None] == classes[None, :], 1.0, -1.0)
    return Y[:, 1:] if classes.size == 2 else Y


def classifier_fit_predict(Xtr, ytr, Xte, alpha) -> np.ndarray:
    classes = np.unique(ytr)
    W, b = ridge_fit(Xtr, binarize(ytr, classes), alpha)
    s = Xte @ W + b
    idx = (s[:, 0] > 0).astype(np.int64) if classes.size == 2 else np.argmax(s, axis=1)
    return classes[idx]


def r2(y: np.ndarray, pred: np.ndarray) -> float:
    """sklearn.metrics.r2_score, multioutput='uniform_average'."""
    y = y.reshape(y.shape[0], -1)
    pred = pred.reshape(y.shape)
    res = ((y - pred) ** 2).sum(axis=0)
    tot = ((y - y.mean(axis=0)) ** 2).sum(axis=0)
    return float(np.mean(1.0 - res / tot))


def probe(train_X: np.ndarray, train_y: np.ndarray, val_X: np.ndarray, val_y: np.ndarray,
          kind: str) -> Dict[str, object]:
    """evaluate.py:36-47 for one task -> {'score', 'alpha', 'cv_scores' (alphas x folds)}."""
    scale, offset = minmax_fit(train_X)
    X = minmax_apply(train_X, scale, offset)
    Xv = minmax_apply(val_X, scale, offset)
    if kind == "regression":
        folds = kfold_ids(X.shape[0], 3)
        Y = train_y.reshape(train_y.shape[0], -1).astype(np.float64)

        def fold_score(tr, te, alpha):
            W, b = ridge_fit(X[tr], Y[tr], alpha)
            return r2(Y[te], X[te] @ W + b)
    else:
        folds = stratified_kfold_ids(train_y, 10)

        def fold_score(tr, te, alpha):
            return float(np.mean(classifier_fit_predict(X[tr], train_y[tr], X[te], alpha)
                                 == train_y[te]))
    k = int(folds.max()) + 1
    cv: List[List[float]] = []
    for alpha in ALPHAS:
        cv.append([fold_score(folds != f, folds == f, alpha) for f in range(k)])
    cv_scores = np.asarray(cv)
    best = int(np.argmax(cv_scores.mean(axis=1)))        # first maximum, like rank 'min' + argmin
    alpha = ALPHAS[best]
    if kind == "regression":
        W, b = ridge_fit(X, Y, alpha)
        score = r2(val_y.reshape(val_y.shape[0], -1).astype(np.float64), Xv @ W + b)
    else:
        score = float(np.mean(classifier_fit_predict(X, train_y, Xv, alpha) == val_y))
    return {"score": score, "alpha": alpha, "cv_scores": cv_scores}
