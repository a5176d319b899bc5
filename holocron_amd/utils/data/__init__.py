from .collate import *  # noqa: F401,F403
