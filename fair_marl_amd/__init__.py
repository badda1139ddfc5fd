"""fair_marl_amd -- MI355X-native rollout hot path of Fair-MARL behind the reference's vec-env API.

The step / observation / reward / reset path runs as hand-written HIP kernels for gfx950
(``csrc/``, C-ABI in ``include/fmarl.h``); this package is the thin Python host side that mirrors
the reference's ``GraphMPEEnv`` + ``onpolicy.envs.env_wrappers`` interface.  There is no CPU
fallback: importing works anywhere, but creating an engine without ``libfmarl.so`` or without a
GPU raises.
"""
from .config import EnvConfig  # noqa: F401
from .engine import OutputRing, RolloutEngine  # noqa: F401
from .env_wrappers import (DummyVecEnv, GraphDummyVecEnv, GraphSubprocVecEnv,  # noqa: F401
                           ShareVecEnv, SubprocVecEnv)
from .MPE_env import GraphMPEEnv, MPEEnv  # noqa: F401
from .pipeline import PipelinedRollout  # noqa: F401
from .rollout_buffer import DeviceRolloutBuffer  # noqa: F401
from .sharding import SpanGather, SpanRecord, StepRecord, TrajectoryGather, shard_range  # noqa: F401
from .spaces import Box, Discrete  # noqa: F401

__version__ = '0.1.0'
