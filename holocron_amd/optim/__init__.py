from .adabelief import *  # noqa: F401,F403
from .lars import *  # noqa: F401,F403
from .adamp import *  # noqa: F401,F403
from .lamb import *  # noqa: F401,F403
from .tadam import *  # noqa: F401,F403
from .wrapper import *  # noqa: F401,F403
