"""sus-net_amd: MI355X-native batched Sus-Net environment (hot path only).

The directory name carries a hyphen, so import it as ``import susnet_amd`` (alias module at the repo root)
or ``importlib.import_module("sus-net_amd")``.
"""
from .metrics import SusMetrics  # noqa: F401
from .env import (  # noqa: F401
    Action, BatchedFourRoomEnv, BatchedFourRoomEnvWithTagging, BatchedImposterTrainingGround, ObsConfig,
    StateFields, four_room_grid,
)
from . import _lib, build_hip, dist, features, policy, replay  # noqa: F401
from .replay import Batch, DeviceReplayBuffer  # noqa: F401
from .policy import MLP, PolicyRollout, RandomEquiprobable, SpatialDQN, WindowedPolicyRollout  # noqa: F401
from .features import FlatFeaturizer, GlobalFeaturizer, PerspectiveFeaturizer  # noqa: F401

# reference names (src/environment/__init__.py:1-3)
FourRoomEnv = BatchedFourRoomEnv
FourRoomEnvWithTagging = BatchedFourRoomEnvWithTagging
ImposterTrainingGround = BatchedImposterTrainingGround
