"""MI355X-native forward path for the vision-MLP model families of liuruiyang98/Jittor-MLP.

The package directory is named `jittor-mlp_amd` (import it with
`importlib.import_module("jittor-mlp_amd")`); `models_pytorch` inside it mirrors the reference's
package of the same name.  Compute runs in libmlpk.so (hand-written gfx950 HIP kernels behind the
C ABI of include/mlpk.h); see DESIGN.md.
"""
from . import _native, engine  # noqa: F401
from . import models_pytorch  # noqa: F401
from .models_pytorch import *  # noqa: F401,F403

__version__ = "0.1.0"
