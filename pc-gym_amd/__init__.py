"""pcgym_amd -- MI355X-native batched process-control environment engine.

Keeps pc-gym's ``make_env(env_params)`` / ``reset()`` / ``step()`` surface
(reference: src/pcgym/__init__.py:1, src/pcgym/pcgym.py:31-500) and runs the
per-timestep hot path as HIP kernels behind the C ABI in include/pcgym_hip.h.
"""
from .config import EnvSpec  # noqa: F401
from .env import StepGraph, VecEnv, make_env, make_vec_env  # noqa: F401
from .gather import HostGather  # noqa: F401
from .mixed import MixedVecEnv, make_mixed_sharded_env, mixed_shard_layout  # noqa: F401
from .reference_engine import hip_integration_engine  # noqa: F401
from .rollout import collect_rollouts, reproducibility_metric  # noqa: F401
from .spaces import Box  # noqa: F401

__version__ = "0.1.0"
