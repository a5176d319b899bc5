from .core import *  # noqa: F401,F403
from .classification import *  # noqa: F401,F403
from .detection import *  # noqa: F401,F403
from .segmentation import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
