from .yolo import *  # noqa: F401,F403
from .yolov2 import *  # noqa: F401,F403
from .yolov4 import *  # noqa: F401,F403
