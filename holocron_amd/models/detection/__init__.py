from .yolov4 import *  # noqa: F401,F403
