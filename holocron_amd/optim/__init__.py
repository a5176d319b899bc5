from .adabelief import *  # noqa: F401,F403
from .lars import *  # noqa: F401,F403
from .adamp import *  # noqa: F401,F403
from .lamb import *  # noqa: F401,F403
from .tadam import *  # noqa: F401,F403
from .wrapper import *  # noqa: F401,F403
from . import adan, ademamix, ralars, wrapper  # noqa: F401,E402  (the reference's module paths)
