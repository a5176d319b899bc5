"""holocron/optim/ralars.py module path; the implementation lives next to LAMB (holocron_amd/optim/lamb.py)."""
from .lamb import RaLars  # noqa: F401

__all__ = ["RaLars"]
