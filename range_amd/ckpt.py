"""SatCLIP checkpoint reader for the RANGE path.

Replaces satclip/load.py:3-18: the reference rebuilds the whole Lightning module (vision tower
included) and returns ``.model.location``; only the hyper-parameters of the location encoder and
the six ``model.location.nnet.*`` tensors matter on this path, so only those are read.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np
import torch


@dataclass
class EncoderParams:
    legendre_polys: int
    hidden: int
    num_hidden_layers: int
    embed_dim: int
    harmonics_calculation: str
    weights: List[np.ndarray]   # float64, torch (out,in) layout; last entry = last_layer
    biases: List[np.ndarray]


_POPPED = ("eval_downstream", "air_temp_data_path", "election_data_path")


class _TolerantPickle:
    """pickle module stand-in for torch.load: classes of packages that are not installed
    (a Lightning checkpoint may reference lightning / pytorch_lightning helper types such as
    ``AttributeDict``) are replaced by plain containers instead of failing the load.  Only used
    after the safe ``weights_only=True`` load has refused the file."""
    import pickle as _p

    class Unpickler(_p.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                if "Dict" in name or "dict" in name or "Namespace" in name:
                    return type(name, (dict,), {"__setstate__": lambda self, st: self.update(st)
                                                if isinstance(st, dict) else None})
                return type(name, (), {"__init__": lambda self, *a, **k: None,
                                       "__setstate__": lambda self, st: None,
                                       "__call__": lambda self, *a, **k: None})

    load = staticmethod(_p.load)
    __name__ = "pickle"


def _load_file(path: str):
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        # Lightning checkpoints may carry non-tensor objects (same trust model as the
        # reference's plain torch.load, satclip/load.py:4)
        try:
            return torch.load(path, map_location="cpu", weights_only=False)
        except (ImportError, AttributeError, ModuleNotFoundError):
            return torch.load(path, map_location="cpu", weights_only=False,
                              pickle_module=_TolerantPickle)


def read_checkpoint(path: str) -> EncoderParams:
    ckpt = _load_file(path)
    hp = dict(ckpt["hyper_parameters"])
    for k in _POPPED:
        hp.pop(k)   # KeyError if absent, exactly like satclip/load.py:5-7
    # defaults of SatCLIPLightningModule.__init__ (satclip/main_old.py:15-37)
    le_type = hp.get("le_type", "grid")
    pe_type = hp.get("pe_type", "siren")
    if le_type != "sphericalharmonics" or pe_type != "siren":
        raise NotImplementedError(
            f"location encoder le_type={le_type!r} pe_type={pe_type!r}: only the "
            "sphericalharmonics + siren encoder of SatCLIP-*-L* checkpoints is implemented")
    mode = hp.get("harmonics_calculation", "analytic")
    if mode not in ("analytic", "closed-form"):
        raise NotImplementedError(f"harmonics_calculation={mode!r}")
    L = int(hp.get("legendre_polys", 16))
    hidden = int(hp.get("capacity", 256))
    n_layers = int(hp.get("num_hidden_layers", 2))
    embed = int(hp.get("embed_dim", 512))
    sd = ckpt["state_dict"]
    pre = "model.location.nnet."
    ws, bs = [], []
    for i in range(n_layers):
        ws.append(sd[f"{pre}layers.{i}.weight"])
        bs.append(sd[f"{pre}layers.{i}.bias"])
    ws.append(sd[f"{pre}last_layer.weight"])
    bs.append(sd[f"{pre}last_layer.bias"])
    if f"{pre}layers.{n_layers}.weight" in sd:
        raise ValueError("state_dict has more hidden layers than hyper_parameters say")
    to_np = lambda t: np.ascontiguousarray(t.detach().to(torch.float64).cpu().numpy())  # .double(), range.py:83-84
    return EncoderParams(L, hidden, n_layers, embed, mode, [to_np(w) for w in ws],
                         [to_np(b) for b in bs])
