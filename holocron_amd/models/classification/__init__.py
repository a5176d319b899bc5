from .repvgg import *  # noqa: F401,F403
