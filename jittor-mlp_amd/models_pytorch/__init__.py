"""Drop-in counterpart of the reference's `models_pytorch` package for the hot-path model
families (export list of the reference: models_pytorch/__init__.py:1-22).  Adding the directory
`jittor-mlp_amd/` to sys.path makes `from models_pytorch import MLPMixerForImageClassification`
resolve here."""
from .g_mlp import gMLPForImageClassification  # noqa: F401
from .res_mlp import ResMLPForImageClassification  # noqa: F401
from .mlp_mixer import MLPMixerForImageClassification  # noqa: F401
from .vip import ViP  # noqa: F401
from .s2_mlp_v1 import S2MLPv1, S2MLPv1_deep, S2MLPv1_wide  # noqa: F401
from .s2_mlp_v2 import S2MLPv2  # noqa: F401
from .conv_mixer import ConvMixer  # noqa: F401
from .as_mlp import AS_MLP  # noqa: F401
from .sparse_mlp import SparseMLP  # noqa: F401  
from .hire_mlp import HireMLP  # noqa: F401  
from .ms_mlp import MS_MLP  # noqa: F401  
from .swin_mlp import SwinMLP  # noqa: F401  
from .cycle_mlp import CycleNet, CycleMLP_B1, CycleMLP_B2, CycleMLP_B3, CycleMLP_B4, CycleMLP_B5  # noqa: F401  
from .utils import Shift  # noqa: F401
# secondary classes the reference lets users import from the sub-modules
from .mlp_mixer import MLPMixer  # noqa: F401
from .g_mlp import gMLP  # noqa: F401
from .res_mlp import ResMLP  # noqa: F401
from .vip import WeightedPermutator, Permutator  # noqa: F401
from .s2_mlp_v2 import S2Block  # noqa: F401
from .cycle_mlp import CycleFC, CycleMLP, CycleBlock  # noqa: F401

__all__ = ["gMLPForImageClassification", "ResMLPForImageClassification", "MLPMixerForImageClassification", "ViP",
           "S2MLPv1", "S2MLPv1_deep", "S2MLPv1_wide", "S2MLPv2", "ConvMixer", "AS_MLP", "SparseMLP", "HireMLP", "MS_MLP", "SwinMLP", "CycleNet",
           "CycleMLP_B1", "CycleMLP_B2", "CycleMLP_B3", "CycleMLP_B4", "CycleMLP_B5", "Shift"]
