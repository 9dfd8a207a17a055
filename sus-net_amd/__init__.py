"""sus-net_amd: MI355X-native batched Sus-Net environment (hot path only).

The directory name carries a hyphen, so import it as ``import susnet_amd`` (alias module at the repo root)
or ``importlib.import_module("sus-net_amd")``.
"""
import os as _os

# Kernel arguments in DEVICE memory instead of host memory (a ROCm runtime knob, read when the HIP runtime initialises):
# every wave's first scalar load of its arguments otherwise crosses the host link, ~3-5 us on each of this path's many
# short launches (measured: 13.7 -> 9.1 us per 8-tick rollout launch, 19.8 -> 16.9 us per step() tick).  setdefault:
# an explicit HIP_FORCE_DEV_KERNARG=0 in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .metrics import EpisodicMetricHandler, SusMetrics  # noqa: F401,E402
from .env import (  # noqa: F401,E402
    Action, BatchedFourRoomEnv, BatchedFourRoomEnvWithTagging, BatchedImposterTrainingGround, ObsConfig,
    StateFields, four_room_grid,
)
from . import _lib, build_hip, dist, features, policy, replay  # noqa: F401,E402
from .replay import Batch, DeviceReplayBuffer  # noqa: F401,E402
from .policy import MLP, PolicyRollout, RandomEquiprobable, SpatialDQN, WindowedPolicyRollout  # noqa: F401,E402
from .features import FlatFeaturizer, GlobalFeaturizer, PerspectiveFeaturizer  # noqa: F401,E402

# reference names (src/environment/__init__.py:1-3)
FourRoomEnv = BatchedFourRoomEnv
FourRoomEnvWithTagging = BatchedFourRoomEnvWithTagging
ImposterTrainingGround = BatchedImposterTrainingGround
