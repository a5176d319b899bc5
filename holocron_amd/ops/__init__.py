from . import conv  # noqa: F401
from . import boxes  # noqa: F401
from .boxes import *  # noqa: F401,F403
