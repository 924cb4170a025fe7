from .binding import LvtError, available, require  # noqa: F401
