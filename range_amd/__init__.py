"""range_amd - MI355X-native engine for the RANGE / RANGE+ retrieval-augmented geo-embedding
forward path of mvrl/RANGE (``load_model(...)(locs)``), its batch driver (``save_embeddings``) and
the downstream ridge probe (``evaluate_npz``).  See DESIGN.md."""
from .load_model import load_model  # noqa: F401
from .range import LocationEncoder  # noqa: F401


def __getattr__(name):
    # the driver and the probe are imported on first use (they pull in their own bindings)
    if name == "save_embeddings":
        from .save import save_embeddings
        return save_embeddings
    if name == "evaluate_npz":
        from .evaluate import evaluate_npz
        return evaluate_npz
    raise AttributeError(f"module 'range_amd' has no attribute {name!r}")


__all__ = ["load_model", "LocationEncoder", "save_embeddings", "evaluate_npz"]
