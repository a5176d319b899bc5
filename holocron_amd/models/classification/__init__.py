from .repvgg import *  # noqa: F401,F403
from .darknet import *  # noqa: F401,F403
from .darknetv2 import *  # noqa: F401,F403
from .darknetv3 import *  # noqa: F401,F403
from .darknetv4 import *  # noqa: F401,F403
from .rexnet import *  # noqa: F401,F403
from .mobileone import *  # noqa: F401,F403
