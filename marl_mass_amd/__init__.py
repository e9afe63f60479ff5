"""Importable alias of the `marl-mass_amd/` package directory (a hyphen is not a legal module name)."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "marl-mass_amd"))
from .vec_env import BatchedMergeEnv, VecMergeEnv, hip_library, reduce_rollout_metrics, shard_range  # noqa: E402,F401
from . import _cabi  # noqa: E402,F401
from .compat import MergeEnvCompat, make, CBFType, safety_layer, cbf_factory  # noqa: E402,F401
