from .tools import pair, check_sizes  # noqa: F401
from .shift import Shift, torch_shift  # noqa: F401
