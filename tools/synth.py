"""Seeded synthetic checkpoints, banks and queries.

The real SatCLIP checkpoint and ``range_db_*.npz`` files are not available offline
(SURVEY.md section 0, fact 5), so every test, golden vector and benchmark uses data made here.
The *formats* follow the reference exactly so that the same loader path is exercised:

* checkpoint: Lightning-style dict ``{'hyper_parameters': {...}, 'state_dict': {...}}`` with the
  location-encoder tensors under ``model.location.nnet.layers.{i}.{weight,bias}`` and
  ``model.location.nnet.last_layer.{weight,bias}`` (reference: satclip/load.py:3-18,
  satclip/location_encoder.py:73-96, satclip/model_old.py:326-330).
* bank: ``np.savez(locs=(N,2) f64 (lon,lat) deg, image_embeddings=(N,1024),
  satclip_embeddings=(N,256))`` (reference: range/generate_db.py:209-214).

Shapes H (capacity) and N are assumptions for the real files.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Tuple

import numpy as np

EMBED_DIM = 256
VALUE_DIM = 1024

# Named synthetic banks (row counts are assumptions).
BANK_ROWS = {"range_db_med": 50_000, "range_db_large": 100_000}


def make_encoder_weights(L: int = 40, hidden: int = 512, embed_dim: int = EMBED_DIM,
                         num_hidden_layers: int = 2, seed: int = 1234) -> Dict[str, np.ndarray]:
    """SIREN-initialised float64 weights, torch ``(out, in)`` layout.

    Init ranges follow satclip/location_encoder.py:137-144 with w0=1: first layer
    U(-1/dim_in, 1/dim_in), later layers U(-sqrt(6/dim_in), sqrt(6/dim_in)).
    Returns ``{'layers.0.weight': ..., 'layers.0.bias': ..., ..., 'last_layer.weight': ...}``.
    """
    rng = np.random.default_rng(seed)
    F = L * L
    out: Dict[str, np.ndarray] = {}
    dim_in = F
    for i in range(num_hidden_layers):
        w_std = (1.0 / dim_in) if i == 0 else math.sqrt(6.0 / dim_in)
        out[f"layers.{i}.weight"] = rng.uniform(-w_std, w_std, size=(hidden, dim_in))
        out[f"layers.{i}.bias"] = rng.uniform(-w_std, w_std, size=(hidden,))
        dim_in = hidden
    w_std = math.sqrt(6.0 / dim_in)
    out["last_layer.weight"] = rng.uniform(-w_std, w_std, size=(embed_dim, dim_in))
    out["last_layer.bias"] = rng.uniform(-w_std, w_std, size=(embed_dim,))
    return out


def default_hparams(L: int = 40, hidden: int = 512, embed_dim: int = EMBED_DIM,
                    num_hidden_layers: int = 2,
                    harmonics_calculation: str = "analytic") -> Dict[str, object]:
    """Hyper-parameters in the shape the reference's checkpoint carries
    (satclip/main_old.py:15-37; the three popped keys of satclip/load.py:5-7 included)."""
    return {
        "embed_dim": embed_dim,
        "image_resolution": 32,
        "vision_layers": 1,
        "vision_width": 64,
        "vision_patch_size": 16,
        "in_channels": 4,
        "le_type": "sphericalharmonics",
        "pe_type": "siren",
        "frequency_num": 16,
        "max_radius": 260,
        "min_radius": 1,
        "legendre_polys": L,
        "harmonics_calculation": harmonics_calculation,
        "sh_embedding_dims": 32,
        "learning_rate": 1e-4,
        "weight_decay": 0.01,
        "num_hidden_layers": num_hidden_layers,
        "capacity": hidden,
        "eval_downstream": False,
        "air_temp_data_path": "",
        "election_data_path": "",
    }


def make_checkpoint(L: int = 40, hidden: int = 512, embed_dim: int = EMBED_DIM,
                    num_hidden_layers: int = 2, seed: int = 1234,
                    harmonics_calculation: str = "analytic") -> Dict[str, object]:
    """A checkpoint dict holding only what the RANGE path reads (no vision tower)."""
    import torch

    w = make_encoder_weights(L, hidden, embed_dim, num_hidden_layers, seed)
    sd = {}
    for k, v in w.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        sd[f"model.location.nnet.{k}"] = t
        sd[f"model.nnet.{k}"] = t  # the reference ckpt carries both aliases
    return {"hyper_parameters": default_hparams(L, hidden, embed_dim, num_hidden_layers,
                                                harmonics_calculation),
            "state_dict": sd}


def write_checkpoint(path: str, **kw) -> str:
    import torch

    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(make_checkpoint(**kw), path)
    return path


def make_bank(n_rows: int, seed: int = 2024, n_clusters: int = 32,
              key_dim: int = EMBED_DIM, value_dim: int = VALUE_DIM
              ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Synthetic bank arrays ``(locs f64 (N,2) lon/lat deg, image_embeddings f32 (N,1024),
    satclip_embeddings f32 (N,256))``: locations uniform on the sphere, keys N(0,1) plus one of
    ``n_clusters`` random centres x3 (so similarities are not all ~0), values N(0,1)."""
    rng = np.random.default_rng(seed)
    lon = rng.uniform(-180.0, 180.0, size=n_rows)
    lat = np.degrees(np.arcsin(rng.uniform(-1.0, 1.0, size=n_rows)))
    locs = np.stack([lon, lat], axis=1).astype(np.float64)
    centres = rng.standard_normal((n_clusters, key_dim)).astype(np.float32)
    which = rng.integers(0, n_clusters, size=n_rows)
    keys = rng.standard_normal((n_rows, key_dim), dtype=np.float32) + 3.0 * centres[which]
    values = rng.standard_normal((n_rows, value_dim), dtype=np.float32)
    return locs, values, keys.astype(np.float32)


def write_bank(path: str, n_rows: int, seed: int = 2024, **kw) -> str:
    """Write the bank with the reference's schema (generate_db.py:212-214, uncompressed savez)."""
    locs, values, keys = make_bank(n_rows, seed, **kw)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, locs=locs, image_embeddings=values, satclip_embeddings=keys)
    return path if path.endswith(".npz") else path + ".npz"


def make_queries(n: int, seed: int = 7, lat_max: float = 45.0, lat_min: float | None = None
                 ) -> np.ndarray:
    """(n,2) float64 (lon, lat) degrees. Default: the gated parity band |lat|<=45
    (SURVEY.md section 8(c)); pass lat_min/lat_max for the polar set."""
    rng = np.random.default_rng(seed)
    lon = rng.uniform(-180.0, 180.0, size=n)
    if lat_min is None:
        lat = rng.uniform(-lat_max, lat_max, size=n)
    else:
        mag = rng.uniform(lat_min, lat_max, size=n)
        lat = mag * rng.choice([-1.0, 1.0], size=n)
    return np.stack([lon, lat], axis=1).astype(np.float64)


# ---- downstream probe tasks (range/utils/evaluate.py reads '<task>_train.npz' / '<task>_val.npz'
# with keys 'embeddings' and 'y', written by range/utils/save.py:38, :58) -------------------------
def make_probe_task(kind: str, n_train: int, n_val: int, dim: int, seed: int, n_classes: int = 0,
                    n_targets: int = 1, noise: float = 0.3,
                    prior_scale: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded stand-in for saved embeddings + labels.  ``kind`` is 'regression' (float64 targets,
    shape (n,) or (n,n_targets)) or 'classification' (int64 labels with a skewed class prior so
    that some classes are rare).  Features have unequal scales and offsets and a low-rank
    correlated part, like real embedding columns."""
    rng = np.random.default_rng(seed)
    n = n_train + n_val
    rank = max(4, dim // 8)
    basis = rng.standard_normal((rank, dim)) / math.sqrt(rank)
    latent = rng.standard_normal((n, rank))
    col_scale = np.exp(rng.uniform(-2.0, 1.0, size=dim))
    col_shift = rng.uniform(-3.0, 3.0, size=dim)
    X = (latent @ basis + 0.5 * rng.standard_normal((n, dim))) * col_scale + col_shift
    if kind == "regression":
        w = rng.standard_normal((rank, n_targets))
        y = latent @ w + noise * math.sqrt(rank) * rng.standard_normal((n, n_targets)) + 5.0
        y = y[:, 0] if n_targets == 1 else y
    elif kind == "classification":
        w = rng.standard_normal((rank, n_classes))
        prior = prior_scale * np.log(rng.dirichlet(np.full(n_classes, 0.6)) + 1e-4)
        score = latent @ w + noise * math.sqrt(rank) * rng.gumbel(size=(n, n_classes)) + prior
        y = np.argmax(score, axis=1).astype(np.int64) * 3 + 7      # labels are not 0..C-1
    else:
        raise ValueError(kind)
    return {"train_embeddings": X[:n_train], "train_y": y[:n_train],
            "val_embeddings": X[n_train:], "val_y": y[n_train:]}


def write_probe_task(embeddings_dir: str, model_name: str, task_name: str,
                     task: Dict[str, np.ndarray]) -> Tuple[str, str]:
    """Write the two npz files where evaluate_npz looks for them (evaluate.py:15-16)."""
    d = os.path.join(embeddings_dir, model_name)
    os.makedirs(d, exist_ok=True)
    tr = os.path.join(d, f"{task_name}_train.npz")
    va = os.path.join(d, f"{task_name}_val.npz")
    np.savez(tr, embeddings=task["train_embeddings"], y=task["train_y"])
    np.savez(va, embeddings=task["val_embeddings"], y=task["val_y"])
    return tr, va


#: probe cases shared by the golden generator, the oracle tests and the GPU parity tests:
#: tag -> (task_name as the reference dispatches on it, make_probe_task kwargs)
PROBE_CASES = {
    "reg_d64": ("elevation", dict(kind="regression", n_train=500, n_val=200, dim=64, seed=101)),
    "reg_d1280": ("population", dict(kind="regression", n_train=4000, n_val=1000, dim=1280,
                                     seed=102)),
    "reg_wide": ("temperature", dict(kind="regression", n_train=900, n_val=300, dim=1280,
                                     seed=103)),
    "reg_two_targets": ("nightlights", dict(kind="regression", n_train=1500, n_val=400, dim=256,
                                            seed=104, n_targets=2)),
    "cls_biome": ("biome", dict(kind="classification", n_train=3000, n_val=1000, dim=256,
                                seed=105, n_classes=14)),
    "cls_country": ("country", dict(kind="classification", n_train=5000, n_val=1200, dim=1280,
                                    seed=106, n_classes=40)),
    "cls_checker": ("checker_5", dict(kind="classification", n_train=1200, n_val=500, dim=128,
                                      seed=107, n_classes=2)),
    "cls_rare": ("ecoregion", dict(kind="classification", n_train=400, n_val=150, dim=96,
                                   seed=109, n_classes=25, prior_scale=2.0)),
    "cls_ocean": ("ocean", dict(kind="classification", n_train=2000, n_val=500, dim=1280,
                                seed=108, n_classes=2)),
}


def write_ylm_source(path: str, L: int) -> str:
    """A ``spherical_harmonics_ylm.py`` in the syntax of the reference's generator
    (spherical_harmonics_generate_ylms.py:37-41: one ``def Yl{l}_m{m}(theta, phi): return <expr>``
    per (l, m), sympy's printing of products and sums), rendered from the regenerated coefficient
    table - a stand-in for the user's generated file in tests of ``load_model(sh_source=)`` and
    ``tools/validate_real.py`` (the real file is absent from the snapshot: SURVEY.md fact 3)."""
    from range_amd import sh_table
    t = sh_table.generate_table(L)

    def poly(terms):
        out = ""
        for n, (c, k) in enumerate(terms):
            mag = repr(abs(float(c)))
            body = mag if k == 0 else (f"{mag}*cos(theta)" if k == 1 else f"{mag}*cos(theta)**{k}")
            out += (("-" if c < 0 else "") + body) if n == 0 else ((" - " if c < 0 else " + ") + body)
        return out

    lines = ["import torch", "from torch import cos, sin", ""]
    for l in range(L):
        for m in range(-l, l + 1):
            i = l * L + abs(m)
            terms = [(float(t.coef[j]), int(t.pow[j])) for j in range(int(t.off[i]), int(t.off[i]) + int(t.cnt[i]))]
            front, kx, p2 = float(t.front[i]), int(t.kx[i]), int(t.p2[i])
            if m == 0:
                expr = poly(terms) if terms else (repr(front) if kx == 0 else
                                                  f"{front!r}*cos(theta)" + (f"**{kx}" if kx > 1 else ""))
            else:
                fac = [repr(front)]
                if p2:
                    fac.append("(" + poly([(float(t.a0[i]), 0), (float(t.a2[i]), 2)]) + ")" + ("" if p2 == 2 else f"**{p2 / 2.0}"))
                if terms:
                    fac.append("(" + poly(terms) + ")")
                am = abs(m)
                fac.append(("cos" if m > 0 else "sin") + ("(phi)" if am == 1 else f"({am}*phi)"))
                if kx:
                    fac.append("cos(theta)" + (f"**{kx}" if kx > 1 else ""))
                expr = "*".join(fac)
            name = f"Yl{l}_m{m}".replace("-", "_minus_")
            lines += ["@torch.jit.script", f"def {name}(theta, phi):", f"    return {expr}", ""]
    with open(path, "w") as f:
        f.write("\n".join(lines))
    return path
