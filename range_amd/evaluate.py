"""Downstream ridge probe on MI355X - the drop-in for the reference's range/utils/evaluate.py.

``evaluate_npz(args)`` has the reference's signature and behaviour (evaluate.py:14-47): it reads
``<embeddings_dir>/<location_model_name>/<task_name>_{train,val}.npz`` (keys ``embeddings``, ``y``,
as written by save_embeddings), min-max scales the embeddings, fits ``RidgeClassifierCV(alphas=
(0.1, 1, 10), cv=10)`` for the classification tasks and ``RidgeCV(alphas=(0.1, 1, 10), cv=3)``
otherwise, and returns the validation score (accuracy / R^2).

What scikit-learn does for those two estimators with an integer ``cv`` is a grid search: one
``Ridge`` (``RidgeClassifier``) per (fold, alpha) on the training part of the fold, scored on the
held-out part, the alpha with the best mean score refitted on all rows.  Here every one of those
fits comes from ONE pass over the scaled embeddings: per-fold Gram statistics (Z^T Z, Z^T T,
column sums) on the float64 matrix cores; the statistics of "all rows but fold f" are differences
of those, so the K x 3 ridge systems are assembled and solved (batched blocked Cholesky) without
touching the embeddings again.  Fold assignment (KFold / StratifiedKFold without shuffling), label
encoding and the arg-max over alphas are index logic and stay on the host.

All arithmetic on the data runs in librange_hip.so (include/range_probe.h); there is no CPU path.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from ._probe_native import ProbeEngine

ALPHAS = (0.1, 1.0, 10.0)                                            # evaluate.py:30, :35
_CLASSIFICATION_TASKS = ("ecoregion", "biome", "country", "ocean")   # evaluate.py:28


def _is_classification(task_name: str) -> bool:
    return task_name in _CLASSIFICATION_TASKS or "checker" in task_name


# ---- fold assignment (host, indices only) --------------------------------------------------------
def kfold_ids(n: int, k: int) -> np.ndarray:
    """KFold(k) without shuffling: k contiguous blocks, the first n % k of them one row longer."""
    if k < 2 or k > n:
        raise ValueError(f"Cannot have number of splits n_splits={k} greater than the number "
                         f"of samples: n_samples={n}." if k > n else "k-fold needs k >= 2")
    edges = (np.arange(k + 1) * (n // k)) + np.minimum(np.arange(k + 1), n % k)
    ids = np.empty(n, dtype=np.int64)
    for f in range(k):
        ids[edges[f]:edges[f + 1]] = f
    return ids


def stratified_kfold_ids(y: np.ndarray, k: int) -> np.ndarray:
    """StratifiedKFold(k) without shuffling.  With classes numbered in order of first appearance,
    the class-sorted label sequence is dealt out cyclically to the k folds; that decides how many
    rows of every class a fold gets.  Within a class the rows are then assigned in their original
    order: the first quota to fold 0, the next to fold 1, and so on."""
    labels, first_pos, inverse = np.unique(y, return_index=True, return_inverse=True)
    appearance_rank = np.empty(labels.size, dtype=np.int64)
    appearance_rank[np.argsort(first_pos)] = np.arange(labels.size)
    cls = appearance_rank[inverse.reshape(-1)]
    sizes = np.bincount(cls, minlength=labels.size)
    if np.all(k > sizes):
        raise ValueError(f"n_splits={k} cannot be greater than the number of members in each class.")
    n, n_cls = cls.shape[0], labels.size
    in_class_order = np.sort(cls)
    # quota[f, c] = rows of class c among positions f, f+k, f+2k, ... of the class-sorted sequence
    quota = np.bincount((np.arange(n) % k) * n_cls + in_class_order,
                        minlength=k * n_cls).reshape(k, n_cls)
    ids = np.empty(n, dtype=np.int64)
    by_class = np.argsort(cls, kind="stable")          # rows of class 0 in order, then class 1, ...
    ids[by_class] = np.repeat(np.tile(np.arange(k), n_cls), quota.T.reshape(-1))
    return ids


# ---- the probe -----------------------------------------------------------------------------------
class RidgeProbe:
    """MinMaxScaler + RidgeCV / RidgeClassifierCV(cv=int) of evaluate.py:30-45 on one GPU."""

    def __init__(self, device="cuda", alphas: Sequence[float] = ALPHAS):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"range_amd runs on MI355X GPUs only (device={device!r}); there is "
                               "no CPU path. Use the reference implementation on CPU.")
        self.engine = ProbeEngine(dev)
        self.device = self.engine.device
        self.alphas = tuple(float(a) for a in alphas)

    def _dev(self, a: np.ndarray, dtype) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(self.device).contiguous()

    def fit_score(self, train_X, train_y, val_X, val_y, classification: bool,
                  cv: Optional[int] = None, timings: Optional[dict] = None) -> Dict[str, object]:
        """Returns {'score': validation score, 'alpha': chosen alpha, 'cv_scores': (alphas, folds)}.
        ``timings`` (a dict) receives wall-clock seconds per stage, synchronising after each."""
        import time
        eng = self.engine
        t_last = [time.perf_counter()]

        def lap(name):
            if timings is not None:
                torch.cuda.synchronize(self.device)
                now = time.perf_counter()
                timings[name] = timings.get(name, 0.0) + now - t_last[0]
                t_last[0] = now
        A = len(self.alphas)
        train_y = np.asarray(train_y)
        val_y = np.asarray(val_y)
        Xtr = self._dev(np.asarray(train_X), torch.float64)
        Xv = self._dev(np.asarray(val_X), torch.float64)
        n, d = Xtr.shape
        if Xv.shape[1] != d or train_y.shape[0] != n or val_y.shape[0] != Xv.shape[0]:
            raise ValueError("inconsistent shapes of embeddings / labels")
        if cv is None:
            cv = 10 if classification else 3                         # evaluate.py:30, :35
        lap("h2d")

        # MinMaxScaler.fit: per-column range on the device, the d-vector bookkeeping on the host
        mn, mx, sm = (t.cpu().numpy() for t in eng.colstats(Xtr))
        rng = mx - mn
        rng[rng < 10 * np.finfo(np.float64).eps] = 1.0               # constant columns
        scale = 1.0 / rng
        offset = 0.0 - mn * scale
        shift = (sm / n) * scale + offset      # ~ column means of the scaled data (any origin works)
        scale_d, offset_d, shift_d = (self._dev(v, torch.float64) for v in (scale, offset, shift))

        # folds -> contiguous row blocks
        if classification:
            classes, code = np.unique(train_y, return_inverse=True)
            code = code.reshape(-1)
            n_cls = int(classes.size)
            if n_cls < 2:
                raise ValueError("classification needs at least two classes")
            ids = stratified_kfold_ids(train_y, cv)
        else:
            ids = kfold_ids(n, cv)
        perm = np.argsort(ids, kind="stable")
        sizes = np.bincount(ids, minlength=cv)
        edges = np.concatenate([[0], np.cumsum(sizes)])
        Z = eng.scale_rows(Xtr, self._dev(perm, torch.int64), scale_d, offset_d, shift_d)
        del Xtr
        lap("scale_and_folds")

        # targets.  Classification: the -1/+1 indicator columns as they are (the decision rule
        # compares scores in those units, and the solver handles any origin through the column
        # sums); regression: centred by the mean over all training rows.
        if classification:
            c, first = (1, 1) if n_cls == 2 else (n_cls, 0)
            counts = np.bincount(code, minlength=n_cls)
            code_d = self._dev(code[perm], torch.int32)
            T = eng.onehot(code_d, c, first, self._dev(np.zeros(c), torch.float64))
        else:
            Y = train_y.reshape(n, -1).astype(np.float64)
            c = Y.shape[1]
            ybar = Y.mean(axis=0)
            T = self._dev((Y - ybar)[perm], torch.float64)

        # one pass over the data: per-fold sufficient statistics.  Slot cv stays zero: "all rows
        # minus nothing" is the refit on the whole training set, solved in the same batch.
        Gf, Bf = eng.empty((cv + 1, d, d)), eng.empty((cv + 1, d, c))
        zsumf, tsumf = eng.empty((cv + 1, d)), eng.empty((cv + 1, c))
        for t_ in (Gf, Bf, zsumf, tsumf):
            t_[cv].zero_()
        for f in range(cv):
            if sizes[f] == 0:
                raise ValueError(f"fold {f} is empty")
            r0, r1 = edges[f], edges[f + 1]
            eng.gram(Z[r0:r1], T[r0:r1], Gf[f], Bf[f], zsumf[f], tsumf[f])
        Gtot, Btot = eng.sum_parts(Gf), eng.sum_parts(Bf)
        zsum, tsum = eng.sum_parts(zsumf), eng.sum_parts(tsumf)
        lap("targets_and_gram")

        # every (fold, alpha) fit and the refit on all rows for every alpha: one batched solve
        W, c0 = eng.solve(Gtot, Btot, zsum, tsum, [float(n - s) for s in sizes] + [float(n)],
                          self.alphas, Gf, Bf, zsumf, tsumf)
        W_all, c0_all = W[cv:cv + 1], c0[cv:cv + 1]
        lap("solve")

        # held-out scores
        cv_scores = np.empty((A, cv))
        for f in range(cv):
            r0, r1 = edges[f], edges[f + 1]
            P = eng.gemm(Z[r0:r1], W[f].view(d, A * c))
            if classification:
                present = self._dev((counts - np.bincount(code[perm[r0:r1]], minlength=n_cls)) > 0,
                                    torch.int32)
                hits = eng.accuracy(P, c0[f], code_d[r0:r1], c, A, n_cls, present)
                cv_scores[:, f] = hits.cpu().numpy() / float(r1 - r0)
            else:
                s = eng.r2_sums(P, c0[f], T[r0:r1], tsumf[f], A).cpu().numpy()
                cv_scores[:, f] = np.mean(1.0 - s[:, :, 0] / s[:, :, 1], axis=1)
        best = int(np.argmax(cv_scores.mean(axis=1)))                # first maximum wins
        lap("fold_scores")

        # validation score of the refit
        Zv = eng.scale_rows(Xv, None, scale_d, offset_d, shift_d)
        Pv = eng.gemm(Zv, W_all[0].view(d, A * c))
        if classification:
            pos = np.searchsorted(classes, val_y.reshape(-1))
            pos[pos == n_cls] = 0
            code_v = np.where(classes[pos] == val_y.reshape(-1), pos, -1)   # unseen label: never hit
            hits = eng.accuracy(Pv, c0_all[0], self._dev(code_v, torch.int32), c, A, n_cls,
                                self._dev(np.ones(n_cls), torch.int32))
            score = float(hits[best].item()) / float(val_y.shape[0])
        else:
            Tv = self._dev(val_y.reshape(val_y.shape[0], -1).astype(np.float64) - ybar,
                           torch.float64)
            _, _, tv_sum = eng.colstats(Tv)
            s = eng.r2_sums(Pv, c0_all[0], Tv, tv_sum, A).cpu().numpy()
            score = float(np.mean(1.0 - s[best, :, 0] / s[best, :, 1]))
        lap("validation")
        return {"score": score, "alpha": self.alphas[best], "cv_scores": cv_scores}


def evaluate_npz(args):
    """Reference: range/utils/evaluate.py:14-47 (same arguments, prints, errors and return)."""
    train_path = os.path.join(args.embeddings_dir, args.location_model_name,
                              args.task_name + "_train.npz")
    val_path = os.path.join(args.embeddings_dir, args.location_model_name,
                            args.task_name + "_val.npz")
    assert os.path.exists(train_path), f"Train embeddings file does not exist: {train_path}"
    assert os.path.exists(val_path), f"Val embeddings file does not exist: {val_path}"
    train_data = np.load(train_path, allow_pickle=False)
    val_data = np.load(val_path, allow_pickle=False)
    if _is_classification(args.task_name):                           # evaluate.py:28-30
        print("Classification Model")
        classification = True
    elif "inat" in args.task_name:
        raise NotImplementedError("Inat evaluation not implemented")   # evaluate.py:31-32
    else:
        print("Regression Model")
        classification = False
    device = getattr(args, "device", "cuda")
    probe = RidgeProbe("cuda" if device in (None, "gpu") else device)
    result = probe.fit_score(train_data["embeddings"], train_data["y"], val_data["embeddings"],
                             val_data["y"], classification)
    val_accuracy = result["score"]
    print(f"The validation set accuracy is {val_accuracy:3f}")
    return val_accuracy
