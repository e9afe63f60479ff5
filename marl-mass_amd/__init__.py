"""MI355X-native batched on-ramp-merge environment + CBF shield (hot path of hkbharath/MARL-MASS)."""
from . import _cabi  # noqa: F401
from .vec_env import BatchedMergeEnv, VecMergeEnv, hip_library, reduce_rollout_metrics, shard_range  # noqa: F401
from .compat import MergeEnvCompat, make, CBFType, safety_layer, cbf_factory  # noqa: F401
