from . import data, metrics  # noqa: F401
