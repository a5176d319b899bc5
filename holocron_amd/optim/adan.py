"""holocron/optim/adan.py module path; the implementation lives next to TAdam (holocron_amd/optim/tadam.py)."""
from .tadam import Adan  # noqa: F401

__all__ = ["Adan"]
