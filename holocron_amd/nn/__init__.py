from . import functional, init  # noqa: F401
from .modules import *  # noqa: F401,F403
