"""range_amd - MI355X-native engine for the RANGE / RANGE+ retrieval-augmented geo-embedding
forward path of mvrl/RANGE (``load_model(...)(locs)``).  See DESIGN.md."""
from .load_model import load_model  # noqa: F401
from .range import LocationEncoder  # noqa: F401

__all__ = ["load_model", "LocationEncoder"]
