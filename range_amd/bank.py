"""Bank (range_db_*.npz) reader and host-side preparation.

Wire format: range/generate_db.py:209-214 ``np.savez(locs=(N,2) f64 (lon,lat) deg,
image_embeddings=(N,1024), satclip_embeddings=(N,256))``.
Preparation follows range/range.py:78-95 in the same dtypes and order, because the trained
behaviour depends on these roundings: locs are cast to float32 BEFORE the trigonometry, keys are
L2-normalised in numpy float32, values are only cast.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass
class PreparedBank:
    keys: np.ndarray     # (N,256) f32, unit rows
    values: np.ndarray   # (N,1024) f32
    xyz: np.ndarray      # (N,3) f32

    @property
    def n_rows(self) -> int:
        return int(self.keys.shape[0])

    def rows(self, start: int, stop: int) -> "PreparedBank":
        return PreparedBank(self.keys[start:stop], self.values[start:stop], self.xyz[start:stop])


def lonlat_rad_to_xyz(rad: np.ndarray) -> np.ndarray:
    """range/utils/utils.py:11-16 (column 0 = lon, column 1 = lat)."""
    cl = np.cos(rad[:, 1])
    return np.stack([cl * np.cos(rad[:, 0]), cl * np.sin(rad[:, 0]), np.sin(rad[:, 1])], axis=1)


def prepare_bank(locs, image_embeddings, satclip_embeddings) -> PreparedBank:
    locs32 = np.asarray(locs).astype(np.float32)                       # range.py:79
    keys = np.asarray(satclip_embeddings).astype(np.float32)           # :85
    if keys.ndim != 2 or keys.shape[1] != 256:
        raise ValueError(f"satclip_embeddings must be (N,256), got {keys.shape}")
    with np.errstate(invalid="ignore", divide="ignore"):
        keys = keys / np.linalg.norm(keys, ord=2, axis=1, keepdims=True)   # :89
    values = np.asarray(image_embeddings).astype(np.float32)           # :90
    if values.shape != (keys.shape[0], 1024):
        raise ValueError(f"image_embeddings must be (N,1024), got {values.shape}")
    if locs32.shape != (keys.shape[0], 2):
        raise ValueError(f"locs must be (N,2), got {locs32.shape}")
    xyz = lonlat_rad_to_xyz(locs32 * math.pi / 180)                    # :93-95 (float32)
    _refuse_degenerate_rows(keys, values, xyz)
    return PreparedBank(np.ascontiguousarray(keys), np.ascontiguousarray(values),
                        np.ascontiguousarray(xyz.astype(np.float32)))


def _refuse_degenerate_rows(keys, values, xyz) -> None:
    """A bank row that is not finite after the reference's preparation - a zero-norm
    ``satclip_embeddings`` row (0/0 at range.py:89), NaN / infinite embeddings or locations - is
    refused at load time, by row number.  The reference loads such a bank without complaint and then
    returns NaN for EVERY query of every batch: the row's logit is NaN and one NaN poisons the whole
    softmax row (range.py:213-215, 231-236), so all 1024 retrieved columns of all queries are NaN.
    That is never what a user wants from a 100 000-row bank with one bad row; the decision here
    (round 6; tests/test_host_cpu.py) is to fail where the cause is, with its row index."""
    for name, a in (("satclip_embeddings (after L2 normalisation: a zero-norm row?)", keys),
                    ("image_embeddings", values), ("locs", xyz)):
        ok = np.isfinite(a).all(axis=1)
        if not ok.all():
            bad = np.flatnonzero(~ok)
            raise ValueError(f"bank row {int(bad[0])} ({bad.size} row(s) in all): {name} not finite. The reference would "
                             "load this bank and return NaN for every query (one NaN logit poisons each softmax row, "
                             "range/range.py:213-215); drop or repair the row(s)")


def load_bank(path: str) -> PreparedBank:
    # the reference uses allow_pickle=True (range.py:78); nothing in the format needs pickle
    with np.load(path, allow_pickle=False) as z:
        return prepare_bank(z["locs"], z["image_embeddings"], z["satclip_embeddings"])
