from .tools import pair, check_sizes  # noqa: F401
