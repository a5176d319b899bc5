"""holocron/optim/ademamix.py module path; the implementation lives next to AdamP (holocron_amd/optim/adamp.py)."""
from .adamp import AdEMAMix  # noqa: F401

__all__ = ["AdEMAMix"]
