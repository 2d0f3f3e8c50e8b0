#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python.

Run in the development container only (needs /root/reference, sympy; never on the GPU box):

    python tests/golden/make_golden.py

What is executed from the reference (read-only, nothing is written under /root/reference, no
bytecode is written, no reference source is copied into this repository):

* ``spherical_harmonics_generate_ylms.py`` is exec'd with its literal ``L = 101`` replaced by
  ``L = 40`` and stdout captured to a temp file.  This regenerates the machine-generated
  ``spherical_harmonics_ylm.py`` that is missing from the snapshot (.MISSING_LARGE_BLOBS:1).
* Level 1 - encoder only, no stand-ins: ``satclip/positional_encoding`` and
  ``satclip/location_encoder.py`` are imported under a private package name (this skips
  ``satclip/__init__.py``, which pulls in ``lightning``), giving the reference's
  ``SphericalHarmonics`` (both ``harmonics_calculation`` modes), ``SirenNet`` and
  ``LocationEncoder``.
* Level 2 - the whole ``load_model(...)`` -> ``LocationEncoder.forward`` path of
  ``range/load_model.py`` / ``range/range.py``.  The modules below are absent from this image and
  are only needed for *base classes or names at import time* (none of their code runs on the
  RANGE path); inert placeholders are put into ``sys.modules`` for them: lightning(.pytorch
  .callbacks/.cli), timm, torchgeo(.models/.datasets.geo), rasterio, torchvision(.transforms),
  albumentations(.core.transforms_interface/.pytorch), geoclip/rshf if requested lazily.

Inputs are synthetic and seeded (tools.synth): fixtures hold the seeds, the query coordinates
and the reference's outputs, not the multi-MB weights/banks, which tests rebuild from the seeds.
"""
from __future__ import annotations

import contextlib
import importlib
import importlib.util
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SAT = os.path.join(REF, "range/location_models/satclip")
PE_DIR = os.path.join(SAT, "positional_encoding")
sys.path.insert(0, REPO)

from tools import synth  # noqa: E402


# ----------------------------------------------------------------------------------------------
def regenerate_ylm(tmp: str, L: int = 40) -> str:
    out = os.path.join(tmp, "spherical_harmonics_ylm.py")
    if os.path.exists(out):
        return out
    src = open(os.path.join(PE_DIR, "spherical_harmonics_generate_ylms.py")).read()
    assert "L = 101" in src
    src = src.replace("L = 101", f"L = {L}")
    buf = io.StringIO()
    argv = sys.argv
    sys.argv = ["spherical_harmonics_generate_ylms.py"]
    try:
        with contextlib.redirect_stdout(buf):
            exec(compile(src, "spherical_harmonics_generate_ylms.py", "exec"), {"__name__": "gen"})
    finally:
        sys.argv = argv
    with open(out, "w") as f:
        f.write(buf.getvalue())
    return out


def import_level1(tmp: str):
    """Reference encoder modules under the private package name ``refsat``."""
    pkg = types.ModuleType("refsat")
    pkg.__path__ = [SAT]
    sys.modules["refsat"] = pkg
    spec = importlib.util.spec_from_file_location(
        "refsat.positional_encoding", os.path.join(PE_DIR, "__init__.py"),
        submodule_search_locations=[PE_DIR, tmp])
    pe = importlib.util.module_from_spec(spec)
    sys.modules["refsat.positional_encoding"] = pe
    spec.loader.exec_module(pe)
    return importlib.import_module("refsat.location_encoder")


def install_placeholders():
    """Inert placeholders for import-time-only dependencies of range/__init__.py (see header)."""
    import torch.nn as nn

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _LM(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    lp = mod("lightning.pytorch", LightningModule=_LM, LightningDataModule=_LM)
    mod("lightning.pytorch.callbacks", ModelCheckpoint=_Any)
    mod("lightning.pytorch.cli", LightningCLI=_Any)
    mod("lightning", pytorch=lp, LightningModule=_LM, LightningDataModule=_LM)
    mod("pytorch_lightning", LightningModule=_LM, LightningDataModule=_LM)
    vt = mod("timm.models.vision_transformer", VisionTransformer=type("VisionTransformer", (), {}))
    tm = mod("timm.models", vision_transformer=vt)
    mod("timm", models=tm, create_model=_Any())
    mod("torchgeo.models", ResNet18_Weights=_Any(), ResNet50_Weights=_Any(),
        ViTSmall16_Weights=_Any())
    geo = mod("torchgeo.datasets.geo", NonGeoDataset=object)
    ds = mod("torchgeo.datasets", geo=geo)
    mod("torchgeo", models=sys.modules["torchgeo.models"], datasets=ds)
    mod("rasterio")
    tvt = mod("torchvision.transforms")
    mod("torchvision", transforms=tvt)
    mod("albumentations.core.transforms_interface", ImageOnlyTransform=object)
    mod("albumentations.core")
    mod("albumentations.pytorch", ToTensorV2=_Any)
    mod("albumentations")
    for name in ("wandb", "h5py", "cv2", "cartopy", "cartopy.crs", "skimage", "skimage.io",
                 "rshf", "rshf.satmae", "geoclip", "open_clip"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                mod(name, **{"SatMAE": _Any, "LocationEncoder": _Any})


def import_level2(tmp: str):
    install_placeholders()
    sys.path.insert(0, REF)
    name = "range.location_models.satclip.positional_encoding.spherical_harmonics_ylm"
    spec = importlib.util.spec_from_file_location(name, os.path.join(tmp, "spherical_harmonics_ylm.py"))
    ylm = importlib.util.module_from_spec(spec)
    sys.modules[name] = ylm
    spec.loader.exec_module(ylm)
    return importlib.import_module("range.load_model")


# ----------------------------------------------------------------------------------------------
def ref_encoder(le, L, hidden, layers, seed, mode):
    posenc = le.get_positional_encoding("sphericalharmonics", legendre_polys=L,
                                        harmonics_calculation=mode).double()
    nnet = le.get_neural_network("siren", input_dim=posenc.embedding_dim, num_classes=256,
                                 dim_hidden=hidden, num_layers=layers).double()
    w = synth.make_encoder_weights(L, hidden, 256, layers, seed)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    nnet.load_state_dict(sd, strict=True)
    return posenc, le.LocationEncoder(posenc, nnet).double().eval()


def main():
    tmp = os.path.join(tempfile.gettempdir(), "range_golden_tmp")
    os.makedirs(tmp, exist_ok=True)
    regenerate_ylm(tmp, 40)
    manifest = {"torch": torch.__version__, "numpy": np.__version__, "cases": []}

    # ---------------- Level 1: SH features + encoder -------------------------------------
    le = import_level1(tmp)
    q_band = synth.make_queries(48, seed=11, lat_max=45.0)
    q_polar = synth.make_queries(16, seed=12, lat_min=45.0, lat_max=89.5)
    q_readme = np.random.default_rng(13).uniform(0.0, 1.0, size=(8, 2))   # Readme.md:86
    q_edge = np.array([[0.0, 0.0], [-180.0, 0.0], [180.0, 0.0], [12.5, -33.0], [0.0, 45.0],
                       [77.0, -45.0], [-123.4, 10.0], [179.999, 44.999]], dtype=np.float64)
    queries = np.concatenate([q_band, q_polar, q_readme, q_edge], axis=0)
    with torch.no_grad():
        for mode in ("analytic", "closed-form"):
            for (L, hidden, layers, seed) in ((40, 512, 2, 1234), (40, 256, 2, 1234),
                                              (10, 64, 2, 5), (16, 128, 3, 6)):
                posenc, enc = ref_encoder(le, L, hidden, layers, seed, mode)
                x = torch.from_numpy(queries)
                feats = posenc(x).numpy()
                emb = enc(x).numpy()
                tag = f"enc_{mode.replace('-', '')}_L{L}_H{hidden}_n{layers}"
                keep = np.arange(0, queries.shape[0], 4)       # SH feature rows kept (size)
                np.savez_compressed(os.path.join(HERE, tag + ".npz"), lonlat=queries,
                                    sh_rows=keep, sh_features=feats[keep], embedding=emb,
                                    L=L, hidden=hidden, num_hidden_layers=layers, seed=seed,
                                    mode=mode)
                manifest["cases"].append(tag)
                print("wrote", tag, feats.shape, emb.shape)

    # ---------------- Level 2: load_model -> forward -------------------------------------
    lm = import_level2(tmp)
    import range.range as rr
    from range.location_models.satclip.main_old import SatCLIPLightningModule

    def ref_ckpt(path, L, hidden, layers, seed, mode):
        hp = synth.default_hparams(L, hidden, 256, layers, mode)
        hp_model = {k: v for k, v in hp.items()
                    if k not in ("eval_downstream", "air_temp_data_path", "election_data_path")}
        module = SatCLIPLightningModule(**hp_model)
        sd = module.state_dict()
        w = synth.make_encoder_weights(L, hidden, 256, layers, seed)
        for k, v in w.items():
            for prefix in ("model.location.nnet.", "model.nnet."):
                assert prefix + k in sd, prefix + k
                sd[prefix + k] = torch.from_numpy(v)
        torch.save({"hyper_parameters": hp, "state_dict": sd}, path)

    cases = [
        # tag, L, hidden, layers, wseed, N, bank seed, B, qseed, lat_max
        ("e2e_L40_H512_N3000", 40, 512, 2, 1234, 3000, 2024, 40, 21, 45.0),
        ("e2e_L40_H256_N1537", 40, 256, 2, 1234, 1537, 2025, 24, 22, 45.0),
        ("e2e_L10_H64_N500", 10, 64, 2, 5, 500, 2026, 33, 23, 45.0),
    ]
    for (tag, L, hidden, layers, wseed, N, bseed, B, qseed, lat_max) in cases:
        ck = os.path.join(tmp, tag + ".ckpt")
        db = os.path.join(tmp, tag + ".npz")
        with contextlib.redirect_stdout(io.StringIO()):
            ref_ckpt(ck, L, hidden, layers, wseed, "analytic")
        synth.write_bank(db, N, bseed)
        q = synth.make_queries(B, seed=qseed, lat_max=lat_max)
        out = {"lonlat": q, "L": L, "hidden": hidden, "num_hidden_layers": layers,
               "weight_seed": wseed, "bank_rows": N, "bank_seed": bseed}
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            m = lm.load_model("RANGE", pretrained_path=ck, device="cpu", db_path=db)
            out["range"] = m(torch.from_numpy(q))
            assert m.location_feature_dim == 1280
            for beta in (0.0, 0.25, 0.5, 0.75, 1.0):
                mp = lm.load_model("RANGE+", pretrained_path=ck, device="cpu", db_path=db,
                                   beta=beta)
                out[f"rangeplus_beta{beta}"] = mp(torch.from_numpy(q))
            # default beta
            mp = lm.load_model("RANGE+", pretrained_path=ck, device="cpu", db_path=db)
            assert mp.args.beta == 0.5 and mp.args.temp == 12.0 and mp.args.geo_temp == 40.0
            # similarity rows the reference's matmuls see (range.py:213, :231), for top-k checks
            e = torch.from_numpy(out["rangeplus_beta0.5"][:, 1024:])
            sim = (e.float() @ mp.db_satclip_embeddings.t()).numpy()
            tv, ti = torch.topk(torch.from_numpy(sim), 16, dim=-1)
            out["sem_topk_val"] = tv.numpy()
            out["sem_topk_idx"] = ti.numpy()
        for k, v in out.items():
            if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape[1] == 1280:
                assert v.dtype == np.float64, (k, v.dtype)
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
        manifest["cases"].append(tag)
        print("wrote", tag, out["range"].shape)

    # error behaviour of the reference's loader (range/load_model.py:31-34, range.py:113-114)
    errs = {}
    for label, fn in (
        ("no_pretrained_path", lambda: lm.load_model("RANGE+", db_path=db)),
        ("no_db_path", lambda: lm.load_model("RANGE+", pretrained_path=ck, device="cpu")),
        ("bad_range_name", lambda: lm.load_model("RANGE++", pretrained_path=ck, device="cpu",
                                                 db_path=db)),
        ("unknown_model", lambda: lm.load_model("NoSuchModel", pretrained_path=ck, device="cpu")),
    ):
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                fn()
            errs[label] = "no error"
        except BaseException as ex:  # noqa: BLE001
            errs[label] = type(ex).__name__
    manifest["errors"] = errs
    print(errs)
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
