from . import data, metrics, misc  # noqa: F401
