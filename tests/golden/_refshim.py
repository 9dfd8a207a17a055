"""In-process import shims so the UNMODIFIED reference env can be imported in the build container.

Used only by ``generate_golden.py`` (container-only; ``/root/reference`` never travels to the GPU box).
The reference needs two things this image lacks:

* ``enum.StrEnum`` (Python >= 3.11; the image has 3.10) -- used by ``src/metrics.py:1``.
* ``gymnasium`` -- used by ``src/environment/base.py:7-8`` for ``Env``, ``spaces`` and ``register``.
  The step/reset dynamics touch it only for ``action_space.n`` (base.py:361) and
  ``flatten/unflatten`` (base.py:230-241), i.e. no arithmetic of the hot path lives there.

Nothing here is reference code: these are minimal stand-ins for third-party modules, written from the
public gymnasium API contract (Discrete.n; flatten = concatenate leaves in Tuple order).
"""
from __future__ import annotations

import enum
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _install_strenum() -> None:
    if hasattr(enum, "StrEnum"):
        return

    class StrEnum(str, enum.Enum):
        def __new__(cls, value):
            obj = str.__new__(cls, value)
            obj._value_ = value
            return obj

        def __str__(self):
            return str(self._value_)

        @staticmethod
        def _generate_next_value_(name, start, count, last_values):
            return name.lower()

    enum.StrEnum = StrEnum


class _Space:
    pass


class Discrete(_Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()


class Box(_Space):
    def __init__(self, low, high, shape, dtype=int):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low)
        self.high = np.full(self.shape, high)


class MultiBinary(_Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = (self.n,)


class Tuple(_Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)


def _leaf_size(space) -> int:
    if isinstance(space, Discrete):
        return space.n
    return int(np.prod(space.shape, dtype=np.int64))


def flatten_space(space):
    if isinstance(space, Tuple):
        n = sum(_leaf_size(s) for s in space.spaces)
        return Box(0, 1, (n,), dtype=np.int64)
    return Box(0, 1, (_leaf_size(space),), dtype=np.int64)


def flatten(space, x):
    if isinstance(space, Tuple):
        parts = [np.asarray(flatten(s, xi)).reshape(-1) for s, xi in zip(space.spaces, x)]
        return np.concatenate(parts) if parts else np.zeros(0)
    if isinstance(space, MultiBinary):
        return np.asarray(x, dtype=np.int8).reshape(-1)
    return np.asarray(x, dtype=np.int64).reshape(-1)


def unflatten(space, x):
    x = np.asarray(x)
    if isinstance(space, Tuple):
        out, off = [], 0
        for s in space.spaces:
            n = _leaf_size(s)
            out.append(unflatten(s, x[off : off + n]))
            off += n
        return tuple(out)
    if isinstance(space, MultiBinary):
        return np.asarray(x, dtype=np.int8).reshape(space.shape)
    return np.asarray(x, dtype=np.int64).reshape(space.shape)


class Env:
    def __init__(self, *a, **k):
        pass

    def reset(self, *a, **k):
        raise NotImplementedError

    def step(self, *a, **k):
        raise NotImplementedError


def install() -> None:
    """Install the shims and put the reference on ``sys.path``. Idempotent."""
    _install_strenum()
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")
        for obj in (Discrete, Box, MultiBinary, Tuple, flatten_space, flatten, unflatten):
            setattr(spaces, obj.__name__, obj)
        envs = types.ModuleType("gymnasium.envs")
        reg = types.ModuleType("gymnasium.envs.registration")
        reg.register = lambda *a, **k: None
        envs.registration = reg
        gym.Env = Env
        gym.spaces = spaces
        gym.envs = envs
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
        sys.modules["gymnasium.envs"] = envs
        sys.modules["gymnasium.envs.registration"] = reg
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
