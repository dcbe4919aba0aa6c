from . import misc  # noqa: F401
