from . import detection, presets, utils  # noqa: F401
from .classification import *  # noqa: F401,F403
from . import classification  # noqa: F401
