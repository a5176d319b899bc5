"""holocron_amd — MI355X-native (gfx950) hot path of frgfm/Holocron.

Mirrors the reference's python surface for the path named in BASELINE.json (models, nn, ops,
optim); the math runs in hand-written HIP kernels behind the C ABI of include/holocron_hip.h.
"""
from . import _lib  # noqa: F401
from . import nn, ops, optim, models, trainer, transforms, utils  # noqa: F401

from .ops.conv import bump_weights_epoch  # noqa: E402,F401  (after writing parameters through .data)
from .ops.conv import flush_deferred_wgrads, set_deferred_wgrads  # noqa: E402,F401  (RepBlock weight gradients are queued during backward)
from ._lib import set_deterministic  # noqa: E402,F401  (bit-reproducible steps: single-writer statistics slots)

__version__ = "0.1.0"
