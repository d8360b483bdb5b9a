"""Drop-in counterpart of the reference's `models_pytorch` package for the hot-path model
families (export list of the reference: models_pytorch/__init__.py:1-22).  Adding the directory
`jittor-mlp_amd/` to sys.path makes `from models_pytorch import MLPMixerForImageClassification`
resolve here."""
from .g_mlp import gMLPForImageClassification  # noqa: F401
from .res_mlp import ResMLPForImageClassification  # noqa: F401
from .mlp_mixer import MLPMixerForImageClassification  # noqa: F401

__all__ = ["gMLPForImageClassification", "ResMLPForImageClassification", "MLPMixerForImageClassification"]
