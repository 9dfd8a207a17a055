"""Batched drop-in for the reference environments, running on one MI355X through libsusnet_hip.so.

Mirrors (same ctor kwargs, method names, argument meaning and error types):

* ``BatchedFourRoomEnv``            <- ``FourRoomEnv``            (reference src/environment/base.py:102-582)
* ``BatchedImposterTrainingGround`` <- ``ImposterTrainingGround`` (reference src/environment/pred_prey.py:20-99)
* ``BatchedFourRoomEnvWithTagging`` <- ``FourRoomEnvWithTagging`` (reference src/environment/tagging.py:9-249)

Every per-environment quantity of the reference gains a leading batch dimension ``B``:
``agent_positions [B, A, 2]``, ``alive_agents [B, A]``, rewards ``[B, A]``, ``done [B]``, ``info[name] [B]``.
PyTorch is used only as the owner of device memory and streams; all dynamics run in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from enum import Enum
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .metrics import SusMetrics

# reference src/environment/base.py:171-193
REFERENCE_WALLS = ((0, 4), (2, 4), (3, 4), (4, 4), (5, 4), (6, 4), (8, 4), (4, 0), (4, 2), (4, 3), (4, 5), (4, 6), (4, 8))


class StateFields(Enum):  # reference src/environment/base.py:36-43
    AGENT_POSITIONS = 0
    ALIVE_AGENTS = 1
    JOB_POSITIONS = 2
    JOB_STATUS = 3
    USED_TAGS = 4
    TAG_COUNTS = 5
    TAG_RESET_COUNT = 6


class Action(Enum):  # reference src/environment/base.py:46-66
    STAY = 0
    UP = 1
    DOWN = 2
    LEFT = 3
    RIGHT = 4
    KILL = 5
    FIX = 6
    SABOTAGE = 7

    @property
    def is_move_action(self):
        return self in (Action.UP, Action.DOWN, Action.LEFT, Action.RIGHT, Action.STAY)

    @property
    def is_job_action(self):
        return self in (Action.KILL, Action.FIX, Action.SABOTAGE)


CREW_ACTIONS = [Action.STAY, Action.UP, Action.DOWN, Action.LEFT, Action.RIGHT, Action.FIX]  # base.py:82-89
IMPOSTER_ACTIONS = [Action.STAY, Action.UP, Action.DOWN, Action.LEFT, Action.RIGHT, Action.SABOTAGE, Action.KILL]  # 91-99
CREW_ACTIONS_SIMPLE = CREW_ACTIONS[:5]  # pred_prey.py:4-10
IMPOSTER_ACTIONS_SIMPLE = CREW_ACTIONS[:5] + [Action.KILL]  # pred_prey.py:12-19


def four_room_grid(n: int = 9, include_walls: bool = True) -> np.ndarray:
    """``grid[i, j]`` True = free.  n == 9 is the reference layout (base.py:171-197).  Other sizes (the
    reference hard-codes 9) use the same transpose-symmetric shape: one wall row and column at index
    ``(n - 1) // 2`` with a door gap two cells in from each end of every arm."""
    g = np.ones((n, n), dtype=bool)
    if not include_walls:
        return g
    if n == 9:
        for i, j in REFERENCE_WALLS:
            g[i, j] = False
        return g
    wall = (n - 1) // 2                                    # 9 -> 4, 14 -> 6
    doors = ((wall - 1) // 2, wall + 1 + (n - 1 - wall) // 2)  # 9 -> (1, 7) as the reference, 14 -> (2, 10)
    for i in range(n):
        if i not in doors:
            g[i, wall] = False
            g[wall, i] = False
    return g


_TORCH_TO_SUS = {torch.uint8: L.U8, torch.int32: L.I32, torch.int64: L.I64}


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


class PackedQNet:
    """Device image of a reference ``MLP`` in the layout ``susnet_qnet_forward`` reads (``SusEnv.qnet_pack``)."""

    def __init__(self, components, cdims, dims, packed, host):
        self.components, self.cdims, self.dims, self.packed, self.host = components, cdims, dims, packed, host
        self.q_buf = {}


class ObsConfig:
    """Which fused observation the kernels write next to every step / reset / rollout tick.

    mode: ``None`` | ``"raw"`` (``flatten_state``) | ``"flat"`` (FlatFeaturizer over ``components``) |
    ``"planes"`` (GlobalFeaturizer) | ``"persp"`` (PerspectiveFeaturizer: every agent's rotated channel order, fused).  dtype ``torch.float32`` reproduces the reference tensors,
    ``torch.uint8`` stores the same integers in a quarter of the bytes.
    """

    def __init__(self, mode: Optional[str] = None, components: Sequence[str] = (), dtype=torch.float32):
        assert mode in (None, "raw", "flat", "planes", "persp"), mode
        assert dtype in (torch.float32, torch.uint8, torch.int8)
        self.mode, self.components, self.dtype = mode, list(components), dtype
        if mode == "flat":
            unknown = [c for c in self.components if c not in L.FLAT_COMPONENTS]
            assert self.components and not unknown, f"unknown flat components {unknown}"

    @property
    def code(self):
        return {None: L.OBS_NONE, "raw": L.OBS_RAW, "flat": L.OBS_FLAT, "planes": L.OBS_PLANES, "persp": L.OBS_PERSP}[self.mode]


class BatchedFourRoomEnv:
    VARIANT = L.VARIANT_BASE
    crew_actions = CREW_ACTIONS
    imposter_actions = IMPOSTER_ACTIONS

    def __init__(
        self,
        n_imposters: int,
        n_crew: int,
        n_jobs: int,
        is_action_order_random=True,
        random_state: Optional[int] = None,
        kill_reward: int = -5,
        complete_job_reward=3,
        sabotage_reward=3,
        time_step_reward: int = 0,
        game_end_reward: int = 10,
        dead_penalty: int = -2,
        shuffle_imposter_index: bool = True,
        debug: bool = False,
        max_time_steps=1000,
        include_walls: bool = True,
        *,
        batch: int = 1,
        device="cuda",
        grid: Optional[np.ndarray] = None,
        grid_size: int = 9,
        auto_reset: bool = False,
        rng: str = "philox",
        seed: int = 0,
        env_id_base: int = 0,
        tape_words: int = 1 << 15,
        obs: Optional[ObsConfig] = None,
        reward_dtype=torch.float32,
        export_state: bool = True,
        check_errors: bool = True,
        _tag_reset_interval: int = 50,
        _vote_reward=3,
    ):
        self._validate_init_args(n_imposters, n_crew, n_jobs)
        assert rng in ("philox", "numpy", "tape")
        self.lib = L.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("sus-net_amd runs on a ROCm device only (device='cuda[:i]')")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.batch = int(batch)
        self.is_action_order_random = is_action_order_random
        self.n_imposters, self.n_crew, self.n_jobs = n_imposters, n_crew, n_jobs
        self.n_agents = n_imposters + n_crew
        self.kill_reward, self.complete_job_reward, self.sabotage_reward = kill_reward, complete_job_reward, sabotage_reward
        self.time_step_reward, self.game_end_reward, self.dead_penalty = time_step_reward, game_end_reward, dead_penalty
        self.shuffle_imposter_index = shuffle_imposter_index
        self.max_time_steps = max_time_steps
        self.auto_reset, self.rng_kind, self.seed, self.env_id_base = auto_reset, rng, int(seed), int(env_id_base)
        self.export_state, self.check_errors = export_state, check_errors
        self.reward_dtype = reward_dtype
        self.tape_words = tape_words
        self.debug = debug

        self.grid = np.asarray(grid, dtype=bool) if grid is not None else four_room_grid(grid_size, include_walls)
        n = self.grid.shape[0]
        assert self.grid.shape == (n, n) and n <= L.MAX_GRID, "square grid up to 16x16"
        self.n_rows = self.n_cols = n
        self.valid_positions = np.argwhere(self.grid)
        self.walls = np.argwhere(~self.grid)
        self.n_imposter_actions = len(self.imposter_actions)
        self.n_crew_actions = len(self.crew_actions)
        self.state_fields = {f: i for i, f in enumerate(
            [StateFields.AGENT_POSITIONS, StateFields.ALIVE_AGENTS, StateFields.JOB_POSITIONS, StateFields.JOB_STATUS])}

        cfg = L.Config()
        cfg.struct_bytes, cfg.abi_version = C.sizeof(L.Config), L.ABI_VERSION
        cfg.variant, cfg.batch = self.VARIANT, self.batch
        cfg.n_imposters, cfg.n_crew, cfg.n_jobs, cfg.grid_n = n_imposters, n_crew, n_jobs, n
        for i in range(n):
            cfg.grid_rows[i] = int(sum(1 << j for j in range(n) if self.grid[i, j]))
        cfg.kill_reward, cfg.complete_job_reward, cfg.sabotage_reward = kill_reward, complete_job_reward, sabotage_reward
        cfg.time_step_reward, cfg.game_end_reward, cfg.dead_penalty = time_step_reward, game_end_reward, dead_penalty
        cfg.vote_reward = _vote_reward
        cfg.max_time_steps = max_time_steps
        cfg.is_action_order_random = int(bool(is_action_order_random))
        cfg.shuffle_imposter_index = int(bool(shuffle_imposter_index))
        cfg.tag_reset_interval = _tag_reset_interval
        cfg.auto_reset = int(bool(auto_reset))
        cfg.rng_mode = L.RNG_PHILOX if rng == "philox" else L.RNG_TAPE
        cfg.seed, cfg.env_id_base, cfg.device = self.seed & (2**64 - 1), self.env_id_base, self.device.index
        self._cfg = cfg
        self._h = C.c_void_p()
        rc = self.lib.susnet_create(C.byref(cfg), C.byref(self._h))
        if rc == L.E_INVALID:
            raise AssertionError(self.lib.susnet_last_error().decode())  # reference ctor asserts (base.py:243-249)
        L.check(rc)
        self._layout = L.Layout()
        L.check(self.lib.susnet_get_layout(self._h, C.byref(self._layout)))
        self.action_space_n = self._layout.action_space_n
        B, A, J = self.batch, self.n_agents, self.n_jobs
        dev = self.device
        with torch.cuda.device(dev):
            self._state = torch.empty(self._layout.state_bytes, dtype=torch.uint8, device=dev)
            L.check(self.lib.susnet_bind_state(self._h, self._state.data_ptr(), self._state.numel(), self._stream()))
        # reference-layout views refreshed by the export kernel
        self.agent_positions = torch.zeros(B, A, 2, dtype=torch.int32, device=dev)
        self.alive_agents = torch.zeros(B, A, dtype=torch.bool, device=dev)
        self.imposter_mask = torch.zeros(B, A, dtype=torch.bool, device=dev)
        self.job_positions = torch.zeros(B, J, 2, dtype=torch.int32, device=dev)
        self.completed_jobs = torch.zeros(B, J, dtype=torch.bool, device=dev)
        self.used_tag_actions = torch.zeros(B, A, dtype=torch.bool, device=dev)
        self.tag_counts = torch.zeros(B, A, dtype=torch.int32, device=dev)
        self._timer = torch.zeros(B, dtype=torch.int32, device=dev)
        self._t = torch.zeros(B, dtype=torch.int32, device=dev)
        self._metrics = torch.zeros(B, L.N_METRICS, dtype=torch.int64, device=dev)
        self._rewards = torch.zeros(A, B, dtype=reward_dtype, device=dev)  # [A][B] in memory; exposed as [B, A]
        self._done = torch.zeros(B, dtype=torch.bool, device=dev)
        self._trunc = torch.zeros(B, dtype=torch.bool, device=dev)
        self._actions = torch.zeros(A, B, dtype=torch.uint8, device=dev)
        self._life_sum = torch.zeros(L.N_LIFETIME, dtype=torch.int64, device=dev)
        self._tape = None
        self.metrics = _MetricsView(self)
        self.obs_config = obs or ObsConfig(None)
        self._obs_spec, self.obs, self.obs_non_spatial = self._make_obs(self.obs_config, 1)
        self.agent_action_map = _ActionMapView(self)
        self._rewards_view = self._rewards.t()  # [B, A] view of the [A][B] buffer
        self._actions_view = self._actions.t()
        io = L.StepIO()
        io.rewards = self._rewards.data_ptr()
        io.rewards_dtype = L.F32 if self._rewards.dtype == torch.float32 else L.F64
        io.rewards_layout = L.LAYOUT_AB
        io.done, io.truncated = self._done.data_ptr(), self._trunc.data_ptr()
        if self._obs_spec is not None:
            io.obs = C.pointer(self._obs_spec)
        self._step_io = io
        if random_state is not None:  # base.py:125-126
            self._reseed(random_state)

    # ------------------------------------------------------------------------------------------
    def _validate_init_args(self, n_imposters, n_crew, n_jobs):  # base.py:243-249
        assert n_imposters > 0, f"Must have at least one imposter. Got {n_imposters}."
        assert n_crew > 0, f"Must have at least one crew member. Got {n_crew}."
        assert n_jobs >= 0, f"Must non-negative jobs. Got {n_jobs}."
        assert n_imposters < n_crew, (
            f"Must be more crew members than imposters. Got {n_imposters} imposters and {n_crew} crew members.")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self):
        """Device guard only when another device is current (the context manager costs microseconds per call)."""
        if torch.cuda.current_device() == self.device.index:
            return _NULL_CTX
        return torch.cuda.device(self.device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self.lib.susnet_destroy(h)
            except Exception:
                pass
            self._h = None

    def _make_obs(self, oc: ObsConfig, ticks: int, rows: Optional[int] = None):
        """Spec + output tensors for ``ticks`` x batch observations (or for ``rows`` free-standing rows)."""
        if oc.mode is None:
            return None, None, None
        spec = L.ObsSpec()
        spec.mode = oc.code
        spec.dtype = L.F32 if oc.dtype == torch.float32 else L.U8
        spec.n_components = len(oc.components)
        for i, c in enumerate(oc.components):
            spec.components[i] = L.FLAT_COMPONENTS[c]
        f1, f2 = C.c_int32(0), C.c_int32(0)
        rc = self.lib.susnet_obs_size(self._h, C.byref(spec), C.byref(f1), C.byref(f2))
        if rc == L.E_INVALID:
            raise ValueError(self.lib.susnet_last_error().decode())
        L.check(rc)
        lead = (rows,) if rows is not None else ((ticks, self.batch) if ticks > 1 else (self.batch,))
        A, N = self.n_agents, self.n_rows
        shape1 = (*lead, A + 2, N, N) if oc.mode == "planes" else (*lead, A, A + 2, N, N) if oc.mode == "persp" else (*lead, f1.value)
        shape2 = (*lead, A, f2.value // A) if oc.mode == "persp" else (*lead, f2.value)
        # (every observation kernel writes every row in full -- rows it rejects as zeros -- so the buffers are not pre-filled:
        # a memset of a 700 MB feature batch cost as much as the kernel that fills it)
        out = torch.empty(shape1, dtype=oc.dtype, device=self.device)
        out2 = torch.empty(shape2, dtype=oc.dtype, device=self.device) if f2.value else None
        spec.out = out.data_ptr()
        spec.out2 = out2.data_ptr() if out2 is not None else None
        return spec, out, out2

    def _obs_ptr(self):
        return C.byref(self._obs_spec) if self._obs_spec is not None else None

    # ---- randomness --------------------------------------------------------------------------
    def _reseed(self, seed):
        """np.random.seed(seed) of the reference (base.py:126,267).

        philox: Philox key = seed, every env's stream restarts at word 0 (env b uses counter env_id_base+b).
        numpy : env b consumes numpy's legacy MT19937 stream for seed ``seed + b`` -- at batch 1 the
                decisions equal the reference's for the same seed.
        """
        if self.rng_kind == "philox":
            self.seed = int(seed)
            with self._on_device():
                L.check(self.lib.susnet_seed(self._h, self.seed & (2**64 - 1), 0, self._stream()))
        elif self.rng_kind == "numpy":
            seeds = [int(seed) + b for b in range(self.batch)] if np.isscalar(seed) else [int(s) for s in seed]
            tape = np.stack([np.random.RandomState(s).randint(0, 2**32, size=self.tape_words, dtype=np.uint32) for s in seeds])
            self.set_tape(tape)
        else:
            raise ValueError("rng='tape': call set_tape(words[B, L]) instead of seeding")

    def set_tape(self, words):
        """Bind raw 32-bit words [B, L] (any integer tensor/array holding values < 2**32); cursors restart at 0."""
        w = torch.as_tensor(np.asarray(words.cpu() if isinstance(words, torch.Tensor) else words).astype(np.uint32).view(np.int32))
        assert w.dim() == 2 and w.shape[0] == self.batch
        self._tape = w.contiguous().to(self.device)
        L.check(self.lib.susnet_bind_tape(self._h, self._tape.data_ptr(), self._tape.shape[1]))
        cur = torch.zeros(self.batch, dtype=torch.int64, device=self.device)
        view = L.StateView()
        view.rng_cursor = cur.data_ptr()
        L.check(self.lib.susnet_import_state(self._h, C.byref(view), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()  # `cur` must outlive the async import

    # ---- reference API -----------------------------------------------------------------------
    @property
    def flattened_state_size(self):  # base.py:230-232
        return self._layout.obs_raw_size

    def flatten_state(self, state):  # base.py:234-235 (batched: [B, S])
        parts = [p.reshape(self.batch, -1).to(torch.int64) for p in state]
        return torch.cat(parts, dim=1)

    def unflatten_state(self, flat):  # base.py:237-241
        flat = torch.as_tensor(flat)
        A, J = self.n_agents, self.n_jobs
        lead = flat.shape[:-1]
        out, k = [], 0
        out.append(flat[..., k:k + 2 * A].reshape(*lead, A, 2)); k += 2 * A
        out.append(flat[..., k:k + A]); k += A
        if J > 0 or self.VARIANT == L.VARIANT_TAGGING:
            out.append(flat[..., k:k + 2 * J].reshape(*lead, J, 2)); k += 2 * J
            out.append(flat[..., k:k + J]); k += J
        if self.VARIANT == L.VARIANT_TAGGING:
            out.append(flat[..., k:k + A]); k += A
            out.append(flat[..., k:k + A]); k += A
            out.append(flat[..., k:k + 1])
        return tuple(out)

    @property
    def imposter_idxs(self):
        """[B, n_imposters] ascending agent indices (the reference keeps numpy's draw order, base.py:274-278;
        only the set matters to the dynamics)."""
        return torch.nonzero(self.imposter_mask)[:, 1].reshape(self.batch, self.n_imposters)

    @property
    def crew_mask(self):
        return ~self.imposter_mask

    @property
    def crew_idxs(self):
        """[B, n_crew] ascending agent indices of the crew (base.py:283)."""
        return torch.nonzero(self.crew_mask)[:, 1].reshape(self.batch, self.n_crew)

    @property
    def agent_rewards(self):
        """Rewards of the last step, [B, A] (base.py:369,387 keep them on the env as well as returning them)."""
        return self._rewards_view

    def compute_state_dims(self, state_field):
        """base.py:565-579 evaluated on the reference's observation_space (base.py:211-228, tagging.py:42-60), its quirk
        included: for a Box it returns ``[high[0] - low[0]] * ndim`` -- a 2x2 tensor of N for the (n, 2) position boxes."""
        N, A, J = self.n_rows, self.n_agents, self.n_jobs
        box2 = torch.tensor([[N, N], [N, N]])
        table = {StateFields.AGENT_POSITIONS: box2, StateFields.ALIVE_AGENTS: torch.tensor([A])}
        if J > 0 or self.VARIANT == L.VARIANT_TAGGING:
            table[StateFields.JOB_POSITIONS] = box2
            table[StateFields.JOB_STATUS] = torch.tensor([J])
        if self.VARIANT == L.VARIANT_TAGGING:
            table[StateFields.USED_TAGS] = torch.tensor([A])
            table[StateFields.TAG_COUNTS] = torch.tensor([A])
            table[StateFields.TAG_RESET_COUNT] = torch.tensor([self.tag_reset_interval - 1])
        if state_field not in table:
            raise IndexError("tuple index out of range")  # the reference indexes observation_space[state_field.value]
        return table[state_field]

    @property
    def t(self):
        return self._t

    def _export(self, full: bool):
        view = L.StateView()
        view.agent_positions = self.agent_positions.data_ptr()
        view.alive_agents = self.alive_agents.data_ptr()
        view.metrics = self._metrics.data_ptr()
        view.t = self._t.data_ptr()
        if self.n_jobs:
            view.completed_jobs = self.completed_jobs.data_ptr()
        if full or self.auto_reset:
            view.imposter_mask = self.imposter_mask.data_ptr()
            if self.n_jobs:
                view.job_positions = self.job_positions.data_ptr()
        if self.VARIANT == L.VARIANT_TAGGING:
            view.used_tag_actions = self.used_tag_actions.data_ptr()
            view.tag_counts = self.tag_counts.data_ptr()
            view.tag_reset_timer = self._timer.data_ptr()
        L.check(self.lib.susnet_export_state(self._h, C.byref(view), self._stream()))

    def refresh_roles(self):
        """Refresh ``imposter_mask`` only (one small launch); needed when ``export_state=False``."""
        view = L.StateView()
        view.imposter_mask = self.imposter_mask.data_ptr()
        with torch.cuda.device(self.device):
            L.check(self.lib.susnet_export_state(self._h, C.byref(view), self._stream()))

    def _state_tuple(self):  # base.py:317-323 / 397-402
        return (self.agent_positions, self.alive_agents,
                *([self.job_positions, self.completed_jobs] if self.n_jobs > 0 else []))

    def reset(self, seed: Optional[int] = None, mask=None, **kwargs) -> Tuple[Tuple, Dict]:
        """base.py:251-324.  ``mask`` [B] bool restricts the reset to some environments."""
        with torch.cuda.device(self.device):
            if seed is not None:
                self._reseed(seed)
            elif self.rng_kind == "numpy" and self._tape is None:
                # like the reference's unseeded global RandomState: fresh OS entropy
                self._reseed(int(np.random.SeedSequence().generate_state(1)[0]) % (2**32 - self.batch))
            m = None
            if mask is not None:
                m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            L.check(self.lib.susnet_reset(self._h, m.data_ptr() if m is not None else None, self._obs_ptr(), self._stream()))
            self.reset_generation = getattr(self, "reset_generation", 0) + 1  # (DeviceReplayBuffer.collect drops its carried window when this moves)
            if self.export_state:
                self._export(full=True)
            if self.check_errors:
                self.poll_errors()
        return self._state_tuple(), self.metrics.get_metrics()

    def sample_actions(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """base.py:326-330: uniform role-valid action index per agent; returns [B, A]."""
        with self._on_device():
            if out is None:
                L.check(self.lib.susnet_sample_actions(self._h, self._actions.data_ptr(), L.U8, L.LAYOUT_AB, self._stream()))
                return self._actions_view  # [A][B] in memory, returned as a [B, A] view
            dtype, layout, buf = self._describe_actions(out, output=True)
            L.check(self.lib.susnet_sample_actions(self._h, buf.data_ptr(), dtype, layout, self._stream()))
        return buf

    @staticmethod
    def _policy_opts(epsilon: float, mask_dead: bool, net_crew: "Optional[PackedQNet]" = None, crew_q_out: Optional[torch.Tensor] = None):
        """susnet_policy_opts pointer (None = greedy, dead agents act like everybody else, a random crew).  ``net_crew``: the crew's packed
        network for the one-kernel tick (``crew_q_out``: where its Q rows go, or None)."""
        if not epsilon and not mask_dead and net_crew is None:
            return None
        o = L.PolicyOpts(float(epsilon), 1 if mask_dead else 0)
        if net_crew is not None:
            o.crew_packed, o.crew_dims, o.crew_n_dims = net_crew.packed.data_ptr(), net_crew.cdims, len(net_crew.dims)
            if crew_q_out is not None:
                assert crew_q_out.dtype == torch.float32 and crew_q_out.is_contiguous() and crew_q_out.shape[-1] == net_crew.dims[-1]
                o.crew_q_out = crew_q_out.data_ptr()
        return C.byref(o)

    def policy_actions(self, q_imposter: torch.Tensor, q_crew: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                       epsilon: float = 0.0, mask_dead: bool = False) -> torch.Tensor:
        """Greedy actions of one tick (visualize.py:547-562): every imposter takes ``argmax(q_imposter[b])``, every crew member
        ``argmax(q_crew[b])`` -- or, with ``q_crew=None``, its uniformly random draw from the action stream (what
        ``sample_actions()`` returns for it).  ``epsilon`` > 0: epsilon-greedy (train.py:355-381) -- with that probability an agent takes
        its random draw instead, decided by a third Philox stream of the handle; ``mask_dead``: dead agents get index 0 (train.py).
        One launch (``susnet_policy_actions``); returns ``out`` (default: an int64 ``[B, A]`` buffer the env keeps)."""
        assert q_imposter.dtype == torch.float32 and tuple(q_imposter.shape) == (self.batch, self.n_imposter_actions) and q_imposter.is_contiguous()
        if q_crew is not None:
            assert q_crew.dtype == torch.float32 and tuple(q_crew.shape) == (self.batch, self.n_crew_actions) and q_crew.is_contiguous()
        if out is None:
            if getattr(self, "_policy_actions_buf", None) is None:
                self._policy_actions_buf = torch.zeros(self.batch, self.n_agents, dtype=torch.int64, device=self.device)
            out = self._policy_actions_buf
        dtype, layout, buf = self._describe_actions(out, output=True)
        with self._on_device():
            L.check(self.lib.susnet_policy_actions(self._h, q_imposter.data_ptr(), q_crew.data_ptr() if q_crew is not None else None,
                                                   self._policy_opts(epsilon, mask_dead), buf.data_ptr(), dtype, layout, self._stream()))
        return buf

    def qnet_pack(self, components: Sequence[str], weights, biases, slopes, into: Optional["PackedQNet"] = None) -> Optional["PackedQNet"]:
        """Pack a reference ``MLP`` (dqn.py:72-108: ``weights[l]`` ``[out, in]`` float32, ``biases[l]``, one PReLU slope per hidden
        layer) for ``qnet_forward``.  Returns ``None`` when the library does not serve this handle / feature layout / layer stack
        (callers then run the torch module and hand its Q rows to ``policy_actions``).  ``into``: re-pack changed weights of the same layer
        stack into an existing image, in place (the trainer's optimizer step / target sync, train.py:402-416)."""
        comps = (C.c_int32 * len(components))(*[L.FLAT_COMPONENTS[k] for k in components])
        w = [np.ascontiguousarray(np.asarray(x, dtype=np.float32)) for x in weights]
        b = [np.ascontiguousarray(np.asarray(x, dtype=np.float32)) for x in biases]
        sl = np.ascontiguousarray(np.asarray(slopes, dtype=np.float32).reshape(-1))
        dims = [int(w[0].shape[1])] + [int(x.shape[0]) for x in w]
        if any(x.ndim != 2 for x in w) or any(w[l].shape[1] != dims[l] for l in range(len(w))) or any(b[l].shape != (dims[l + 1],) for l in range(len(w))) \
                or sl.shape[0] != len(w) - 1:
            return None
        cdims = (C.c_int32 * len(dims))(*dims)
        n = self.lib.susnet_qnet_packed_floats(self._h, comps, len(components), cdims, len(dims))
        if n < 0:
            return None
        wp = (C.c_void_p * len(w))(*[x.ctypes.data for x in w])
        bp = (C.c_void_p * len(b))(*[x.ctypes.data for x in b])
        if into is not None:  # new weights into an existing image: same device buffer, so captured graphs that read it stay valid
            assert list(into.dims) == dims and int(n) == into.host.size, "qnet_pack(into=...): another layer stack"
            L.check(self.lib.susnet_qnet_pack(self._h, comps, len(components), cdims, len(dims), wp, bp, sl.ctypes.data, into.host.ctypes.data))
            into.packed.copy_(torch.from_numpy(into.host), non_blocking=False)
            return into
        host = np.empty(int(n), dtype=np.float32)
        L.check(self.lib.susnet_qnet_pack(self._h, comps, len(components), cdims, len(dims), wp, bp, sl.ctypes.data, host.ctypes.data))
        return PackedQNet(comps, cdims, dims, torch.from_numpy(host).to(self.device), host)

    def qnet_forward(self, net: "PackedQNet", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``MLP.forward`` on the CURRENT environments' flat features as one kernel (``susnet_qnet_forward``): ``[B, n_out]`` float32.
        Reads the state, not ``env.obs``: neither the observation nor any activation goes through memory."""
        if out is None:
            out = net.q_buf.get(self.batch)
            if out is None:
                out = net.q_buf[self.batch] = torch.empty(self.batch, net.dims[-1], dtype=torch.float32, device=self.device)
        assert out.dtype == torch.float32 and tuple(out.shape) == (self.batch, net.dims[-1]) and out.is_contiguous()
        with self._on_device():
            L.check(self.lib.susnet_qnet_forward(self._h, net.components, len(net.components), net.cdims, len(net.dims),
                                                 net.packed.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def _describe_actions(self, a: torch.Tensor, output: bool = False):
        """(dtype, layout, tensor) of an action tensor as the C ABI takes it.  An INPUT of another dtype / stride pattern is converted;
        an ``output=True`` buffer must be usable as it is (a converted copy would receive the kernel's writes instead of the caller's
        tensor)."""
        if a is self._actions_view:
            return L.U8, L.LAYOUT_AB, a
        A, B = self.n_agents, self.batch
        if output:
            assert a.dtype in _TORCH_TO_SUS and a.device == self.device, f"action output buffers are uint8 / int32 / int64 on {self.device}"
            assert (tuple(a.shape) == (B, A) and (a.is_contiguous() or a.t().is_contiguous())) or (tuple(a.shape) == (A, B) and a.is_contiguous() and A != B), \
                f"action output buffers are [B, A] (either memory order) or contiguous [A, B], got shape {tuple(a.shape)} stride {a.stride()}"
        if a.dtype not in _TORCH_TO_SUS:
            a = a.to(torch.int64)
        if tuple(a.shape) == (B, A) and a.is_contiguous():
            return _TORCH_TO_SUS[a.dtype], L.LAYOUT_BA, a
        if tuple(a.shape) == (B, A) and a.t().is_contiguous():
            return _TORCH_TO_SUS[a.dtype], L.LAYOUT_AB, a
        if tuple(a.shape) == (A, B) and a.is_contiguous() and A != B:
            return _TORCH_TO_SUS[a.dtype], L.LAYOUT_AB, a
        if tuple(a.shape) == (B, A):
            a = a.contiguous()
            return _TORCH_TO_SUS[a.dtype], L.LAYOUT_BA, a
        raise AssertionError(f"Expected {A} actions per environment, got shape {tuple(a.shape)}")  # base.py:357-359

    def step(self, agent_actions):
        """base.py:332-407.  Returns ``(state, rewards[B, A], done[B], truncated[B], info)``."""
        a = agent_actions
        if not isinstance(a, torch.Tensor):
            a = torch.as_tensor(np.asarray(a))
        if a.dim() == 1 and self.batch == 1:
            a = a.reshape(1, -1)
        assert a.shape[-1] == self.n_agents or tuple(a.shape) == (self.n_agents, self.batch), (
            f"Expected {self.n_agents} actions, got {a.shape[-1]}")  # base.py:357-359
        if a.device != self.device:
            a = a.to(self.device)
        dtype, layout, a = self._describe_actions(a)
        io = self._step_io
        io.actions, io.actions_dtype, io.actions_layout = a.data_ptr(), dtype, layout
        with self._on_device():
            L.check(self.lib.susnet_step(self._h, C.byref(io), self._stream()))
            if self.export_state:
                self._export(full=False)
            if self.check_errors:
                self.poll_errors()
        return self._state_tuple(), self._rewards_view, self._done, self._trunc, self.metrics.get_metrics()

    def policy_step(self, q_imposter: torch.Tensor, q_crew: Optional[torch.Tensor] = None, actions_out: Optional[torch.Tensor] = None,
                    epsilon: float = 0.0, mask_dead: bool = False):
        """``policy_actions`` + ``step`` in ONE launch (``susnet_policy_step``): the stepping lane takes the teams' greedy actions from
        the Q rows itself.  Returns ``(state, rewards, done, truncated, info, actions)``; ``actions`` = ``actions_out`` (default: the
        env's int64 ``[B, A]`` buffer) holding the actions taken."""
        assert q_imposter.dtype == torch.float32 and tuple(q_imposter.shape) == (self.batch, self.n_imposter_actions) and q_imposter.is_contiguous()
        if q_crew is not None:
            assert q_crew.dtype == torch.float32 and tuple(q_crew.shape) == (self.batch, self.n_crew_actions) and q_crew.is_contiguous()
        if actions_out is None:
            if getattr(self, "_policy_actions_buf", None) is None:
                self._policy_actions_buf = torch.zeros(self.batch, self.n_agents, dtype=torch.int64, device=self.device)
            actions_out = self._policy_actions_buf
        dtype, layout, buf = self._describe_actions(actions_out, output=True)
        io = self._step_io
        io.actions, io.actions_dtype, io.actions_layout = buf.data_ptr(), dtype, layout
        with self._on_device():
            L.check(self.lib.susnet_policy_step(self._h, q_imposter.data_ptr(), q_crew.data_ptr() if q_crew is not None else None,
                                                self._policy_opts(epsilon, mask_dead), C.byref(io), self._stream()))
            if self.export_state:
                self._export(full=False)
            if self.check_errors:
                self.poll_errors()
        return self._state_tuple(), self._rewards_view, self._done, self._trunc, self.metrics.get_metrics(), buf

    def supports_qnet_policy_step(self, net: "PackedQNet", net_crew: "Optional[PackedQNet]" = None, epsilon: float = 0.0) -> bool:
        """Whether ``qnet_policy_step`` serves this env: the compiled-in 1v1 9x9 ITG / 1v2 14x14 4-job games (a packed record layout exists
        exactly for the compiled-in games, and among them the agent count names the game) on the production stream -- or, with the crew's
        network too (``net_crew``) and no exploration, on numpy tapes as well: nothing is drawn then."""
        if net is None or net.dims[-1] != self.n_imposter_actions:
            return False
        if net_crew is not None and (net_crew.dims[-1] != self.n_crew_actions or list(net_crew.components) != list(net.components)):
            return False
        if self.rng_kind != "philox" and (net_crew is None or epsilon > 0.0):
            return False
        lay = self.record_layout()
        if lay is None or lay.n_obs_segments != 1:  # (one observation segment: a fully compiled-in game, not a family kernel)
            return False
        return (self.VARIANT, self.n_agents, self.n_jobs, self.n_rows) in ((L.VARIANT_ITG, 2, 0, 9), (L.VARIANT_BASE, 3, 4, 14))

    def qnet_policy_step(self, net: "PackedQNet", actions_out: Optional[torch.Tensor] = None, q_out: Optional[torch.Tensor] = None,
                         epsilon: float = 0.0, mask_dead: bool = False, net_crew: "Optional[PackedQNet]" = None, q_crew_out: Optional[torch.Tensor] = None):
        """A whole tick of the acting loop in ONE kernel (``susnet_qnet_policy_step``): the imposters' network (``net``), its argmax, the
        crew's random draws -- or the crew's own network, ``net_crew`` -- and the step.  Returns like ``policy_step``; raises ``RuntimeError``
        where the library does not serve the configuration (callers fall back to ``qnet_forward`` + ``policy_step``)."""
        if actions_out is None:
            if getattr(self, "_policy_actions_buf", None) is None:
                self._policy_actions_buf = torch.zeros(self.batch, self.n_agents, dtype=torch.int64, device=self.device)
            actions_out = self._policy_actions_buf
        dtype, layout, buf = self._describe_actions(actions_out, output=True)
        if q_out is not None:
            assert q_out.dtype == torch.float32 and tuple(q_out.shape) == (self.batch, net.dims[-1]) and q_out.is_contiguous()
        io = self._step_io
        io.actions, io.actions_dtype, io.actions_layout = buf.data_ptr(), dtype, layout
        with self._on_device():
            L.check(self.lib.susnet_qnet_policy_step(self._h, net.components, len(net.components), net.cdims, len(net.dims), net.packed.data_ptr(),
                                                     q_out.data_ptr() if q_out is not None else None, self._policy_opts(epsilon, mask_dead, net_crew, q_crew_out),
                                                     C.byref(io), self._stream()))
            if self.export_state:
                self._export(full=False)
            if self.check_errors:
                self.poll_errors()
        return self._state_tuple(), self._rewards_view, self._done, self._trunc, self.metrics.get_metrics(), buf

    # ---- policy-driven collection: the trainer's acting loop (train.py:345-399) writing the replay feed tick by tick -------------------
    def alloc_feed(self, n_ticks: int) -> Dict[str, torch.Tensor]:
        """The ``[T][B]`` trajectory block a policy-driven collection of ``n_ticks`` ticks fills, one tick per ``policy_tick_into`` call,
        and ``susnet_ring_append`` then consumes (same tensors a fused rollout with ``replay_feed=True`` writes)."""
        T, B, A, S = int(n_ticks), self.batch, self.n_agents, self.flattened_state_size
        z = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=self.device)
        return {"n_ticks": T, "actions": z(T, B, A, dtype=torch.uint8), "rewards": z(T, B, A, dtype=torch.float32),
                "done": z(T, B, dtype=torch.bool), "truncated": z(T, B, dtype=torch.bool), "obs": z(T, B, S, dtype=torch.uint8),
                "term_obs": z(T, B, S, dtype=torch.uint8), "roles": z(T, B, dtype=torch.int16)}

    def policy_tick_into(self, feed: Dict[str, torch.Tensor], t: int, net_imposter: "PackedQNet" = None, net_crew: "PackedQNet" = None,
                         q_imposter: Optional[torch.Tensor] = None, q_crew: Optional[torch.Tensor] = None, epsilon: float = 0.0,
                         mask_dead: bool = True, q_out: Optional[torch.Tensor] = None) -> None:
        """ONE tick of the trainer's acting loop (train.py:345-399: act on the current state, step, keep what ``replay_buffer.add``
        needs) written into slot ``t`` of ``feed``: the actions taken, rewards, done / truncated, the raw uint8 state after the step (after
        the auto-reset where the episode ended), the true terminal state there, and the acting episode's roles.  The teams' Q rows come
        from packed reference MLPs (``net_*``: the Q-network kernel reads the state itself) or from the caller (``q_*``).  With
        ``net_imposter`` alone on a compiled-in game and the production stream the whole tick is ONE kernel
        (``susnet_qnet_policy_step``: network, argmax / exploration, the random crew's draws, step, feed); else the network launch(es) +
        ``susnet_policy_step``.  Asynchronous; nothing is exported to the host-side mirrors."""
        assert self.auto_reset, "collection needs an auto-resetting env"
        A, S = self.n_agents, self.flattened_state_size
        io = feed.get("_io")
        if io is None:
            io = feed["_io"] = L.StepIO()
            spec = L.ObsSpec()
            spec.mode, spec.dtype = L.OBS_RAW, L.U8
            feed["_spec"] = spec
            io.obs = C.pointer(spec)
            io.actions_dtype, io.actions_layout = L.U8, L.LAYOUT_BA
            io.rewards_dtype, io.rewards_layout = L.F32, L.LAYOUT_BA
        t = int(t)
        # (the observation writer stores 16-byte pieces: a slot that does not start on a 16-byte boundary -- odd small batches -- is
        # written through an aligned bounce buffer)
        slot = feed["obs"][t]
        bounce = None
        if slot.data_ptr() % 16:
            bounce = feed.get("_bounce")
            if bounce is None:
                bounce = feed["_bounce"] = torch.zeros_like(slot)
        feed["_spec"].out = (bounce if bounce is not None else slot).data_ptr()
        io.actions, io.rewards = feed["actions"][t].data_ptr(), feed["rewards"][t].data_ptr()
        io.done, io.truncated = feed["done"][t].data_ptr(), feed["truncated"][t].data_ptr()
        io.term_obs, io.roles = feed["term_obs"][t].data_ptr(), feed["roles"][t].data_ptr()
        one_kernel = (net_imposter is not None and q_imposter is None and q_crew is None and
                      self.supports_qnet_policy_step(net_imposter, net_crew, epsilon))
        opts = self._policy_opts(epsilon, mask_dead, net_crew if one_kernel else None)
        with self._on_device():
            if one_kernel:
                if q_out is not None:  # (the one-kernel tick can also emit the Q rows it acted on)
                    assert q_out.dtype == torch.float32 and tuple(q_out.shape) == (self.batch, net_imposter.dims[-1]) and q_out.is_contiguous()
                L.check(self.lib.susnet_qnet_policy_step(self._h, net_imposter.components, len(net_imposter.components), net_imposter.cdims,
                                                         len(net_imposter.dims), net_imposter.packed.data_ptr(), q_out.data_ptr() if q_out is not None else None,
                                                         opts, C.byref(io), self._stream()))
            else:
                if q_imposter is None:
                    q_imposter = self.qnet_forward(net_imposter)
                if q_crew is None and net_crew is not None:
                    q_crew = self.qnet_forward(net_crew)
                L.check(self.lib.susnet_policy_step(self._h, q_imposter.data_ptr(), q_crew.data_ptr() if q_crew is not None else None, opts,
                                                    C.byref(io), self._stream()))
            if bounce is not None:
                slot.copy_(bounce)

    def policy_rollout_into(self, feed: Dict[str, torch.Tensor], n_ticks: int, net_imposter: "PackedQNet", epsilon: float = 0.0, mask_dead: bool = True,
                            q_out: Optional[torch.Tensor] = None, net_crew: "Optional[PackedQNet]" = None, q_crew_out: Optional[torch.Tensor] = None) -> None:
        """``n_ticks`` policy ticks (slots 0 .. n_ticks - 1 of ``feed``) in ONE launch (``susnet_qnet_policy_rollout``): what ``policy_tick_into``
        does per tick, with the network image loaded and the launch paid once -- the acting loop between two optimizer steps.  Needs
        ``supports_qnet_policy_step(net_imposter)`` (a compiled-in game, the production stream, a random crew)."""
        assert self.auto_reset and self.supports_qnet_policy_step(net_imposter, net_crew, epsilon) and 1 <= n_ticks <= feed["n_ticks"]
        assert feed["obs"][0].data_ptr() % 16 == 0 and (n_ticks == 1 or (self.batch * self.flattened_state_size) % 16 == 0), \
            "the fused raw observation is written in 16-byte pieces: batch x flattened_state_size must be a multiple of 16"
        io = L.FeedIO()
        io.actions, io.rewards = feed["actions"].data_ptr(), feed["rewards"].data_ptr()
        io.done, io.truncated = feed["done"].data_ptr(), feed["truncated"].data_ptr()
        io.obs, io.term_obs, io.roles = feed["obs"].data_ptr(), feed["term_obs"].data_ptr(), feed["roles"].data_ptr()
        if q_out is not None:
            assert q_out.dtype == torch.float32 and tuple(q_out.shape) == (n_ticks, self.batch, net_imposter.dims[-1]) and q_out.is_contiguous()
            io.q = q_out.data_ptr()
        with self._on_device():
            L.check(self.lib.susnet_qnet_policy_rollout(self._h, net_imposter.components, len(net_imposter.components), net_imposter.cdims,
                                                        len(net_imposter.dims), net_imposter.packed.data_ptr(),
                                                        self._policy_opts(epsilon, mask_dead, net_crew, q_crew_out), C.byref(io), int(n_ticks), self._stream()))

    def policy_block(self, n_ticks: int, net_imposter: "PackedQNet", epsilon: float = 0.0, mask_dead: bool = False,
                     net_crew: "Optional[PackedQNet]" = None) -> None:
        """``n_ticks`` ticks of the acting loop in ONE launch with nothing kept but the state and the episode metrics: ``run_game``'s loop with
        fixed networks (visualize.py:547-582), i.e. ``policy_rollout_into`` without a feed.  The fused observation (``env.obs``) is refreshed
        once, after the block."""
        assert self.auto_reset and self.supports_qnet_policy_step(net_imposter, net_crew, epsilon) and n_ticks >= 1
        io = L.FeedIO()
        with self._on_device():
            L.check(self.lib.susnet_qnet_policy_rollout(self._h, net_imposter.components, len(net_imposter.components), net_imposter.cdims,
                                                        len(net_imposter.dims), net_imposter.packed.data_ptr(), self._policy_opts(epsilon, mask_dead, net_crew),
                                                        C.byref(io), int(n_ticks), self._stream()))
            if self._obs_spec is not None:
                L.check(self.lib.susnet_observe(self._h, C.byref(self._obs_spec), self._stream()))
            # once per block: what step() / policy_step() do per tick for a handle built with these flags
            if self.export_state:
                self._export(full=False)
            if self.check_errors:
                self.poll_errors()

    def step4(self, agent_actions):
        """North-star surface ``(obs, rewards, dones, info)``; dones = done | truncated."""
        state, rew, done, trunc, info = self.step(agent_actions)
        obs = self.obs if self._obs_spec is not None else state
        return obs, rew, done | trunc, info

    def native_layout(self):
        """susnet_layout of the handle: sizes, environments per wave of the fused rollout, and which test hooks
        (SUSNET_OVERRIDE_* bits) were found in the environment when it was created."""
        return self._layout

    def set_launch_limit(self, nbytes: int = 0):
        """Largest output array one fused-rollout launch may address (susnet_set_launch_limit; 0 = the default, 2^31 - 1):
        longer trajectories run as consecutive launches.  Launch plumbing only -- results do not depend on it."""
        L.check(self.lib.susnet_set_launch_limit(self._h, int(nbytes)))

    @staticmethod
    def _record_format(packed) -> int:
        """``packed`` of ``alloc_rollout`` / ``rollout``: True = the handle's default record, "compact" = SUSNET_RECORD_COMPACT."""
        return L.RECORD_COMPACT if packed == "compact" else L.RECORD_DEFAULT

    def record_layout(self, packed=True):
        """Field offsets of the packed per-env-step trajectory record (``packed="compact"``: the 16-byte record of the 1v1 no-walls
        game), or None when the configuration has none."""
        lay = L.RecordLayout()
        L.check(self.lib.susnet_record_layout_of(self._h, self._record_format(packed), C.byref(lay)))
        return lay if lay.record_bytes else None

    @staticmethod
    def record_pieces(record_bytes: int):
        """``[(start, width), ...]``: how a ``planar`` packed record is cut (susnet_record_layout_t: 16-byte pieces, then an 8- and / or a
        4-byte piece).  Piece ``(s, w)`` of tick ``t`` is the ``[B][w]`` byte array at ``t * B * R + B * s``."""
        full = record_bytes // 16 * 16
        pieces = [(s, 16) for s in range(0, full, 16)]
        rem = record_bytes - full
        if rem >= 8:
            pieces.append((full, 8))
        if rem in (4, 12):
            pieces.append((full + (8 if rem == 12 else 0), 4))
        return pieces

    def unpack_record(self, record: torch.Tensor, packed=True) -> torch.Tensor:
        """uint8 ``[T, B, record_bytes]`` records in field order (``susnet_record_layout_t`` offsets) from the buffer a packed rollout
        wrote: the buffer itself where the handle stores whole records, a gathered COPY where it stores them as planes of 16-byte pieces
        (``planar``: the multi-agent kernels)."""
        lay = self.record_layout(packed)
        if not lay.planar:
            return record
        T, B, R = record.shape
        flat = record.reshape(T, B * R)
        out = torch.empty_like(record)
        for s, w in self.record_pieces(R):
            out[:, :, s:s + w] = flat[:, B * s:B * s + B * w].reshape(T, B, w)
        return out

    def record_fields(self, record: torch.Tensor, packed=True) -> Dict[str, torch.Tensor]:
        """``actions / rewards / done / truncated / obs`` of a packed record buffer, shaped like the separate trajectory tensors: strided
        VIEWS into the buffer for whole-record layouts, slices of the unpacked copy for planar ones; decoded COPIES of actions and flags
        where they share a byte (``flags_packed``: the compact 1v1 record)."""
        lay, A, F = self.record_layout(packed), self.n_agents, self.flattened_state_size
        rec = self.unpack_record(record, packed)
        if lay.flags_packed:  # one byte: a0 | a1 << 3 | done << 6 | truncated << 7
            fb = rec[:, :, lay.off_actions]
            return {"rewards": rec[:, :, lay.off_rewards:lay.off_rewards + 4 * A].view(torch.float32),
                    "actions": torch.stack([(fb >> (3 * i)) & 7 for i in range(A)], dim=2), "done": ((fb >> 6) & 1).to(torch.bool),
                    "truncated": (fb >> 7).to(torch.bool), "obs": rec[:, :, lay.off_obs:lay.off_obs + F]}
        return {"rewards": rec[:, :, lay.off_rewards:lay.off_rewards + 4 * A].view(torch.float32) if not lay.planar else
                           rec[:, :, lay.off_rewards:lay.off_rewards + 4 * A].contiguous().view(torch.float32),
                "actions": rec[:, :, lay.off_actions:lay.off_actions + A],
                "done": rec[:, :, lay.off_done].view(torch.bool), "truncated": rec[:, :, lay.off_truncated].view(torch.bool),
                "obs": (rec[:, :, lay.off_obs:lay.off_obs + F] if lay.n_obs_segments == 1 else  # (several segments: the family kernels' records)
                        torch.cat([rec[:, :, lay.obs_segments[k][0]:lay.obs_segments[k][0] + lay.obs_segments[k][1]] for k in range(lay.n_obs_segments)], dim=2))}

    def alloc_rollout(self, n_ticks: int, store=("actions", "rewards", "done", "truncated"), obs: Optional[ObsConfig] = None,
                      packed: bool = False, replay_feed: bool = False):
        """Allocate (once) the trajectory buffers a fused rollout of up to ``n_ticks`` ticks writes:
        actions u8 [T, B, A], rewards f32 [T, B, A], done / truncated bool [T, B], obs [T, B, ...].

        ``packed=True`` (compiled-in configurations, full trajectory + raw uint8 observation only): ONE buffer
        ``record`` u8 [T, B, record_bytes] holding the same fields per env-step, which a lane writes with a few wide
        stores; for whole-record layouts the returned ``actions / rewards / done / truncated / obs`` are strided VIEWS into it
        (same shapes and dtypes as the separate tensors); the multi-agent kernels store the record as planes of 16-byte pieces
        (``record_layout().planar``): ``record_fields`` / ``unpack_record`` gather it after a launch.

        ``replay_feed=True`` (full trajectory + raw uint8 observation): also ``term_obs`` u8 [T, B, S] -- written only where
        an episode ended: its true terminal state -- and ``roles`` int16 [T, B] (imposter bitmask of the acting episode): what
        ``DeviceReplayBuffer.populate_fused`` needs besides the trajectory."""
        T, A, B = int(n_ticks), self.n_agents, self.batch
        out = {"n_ticks": T}
        if packed:
            lay = self.record_layout(packed)
            assert lay is not None, "this configuration has no packed record mode" + (" of the compact format" if packed == "compact" else "")
            out["_record_format"] = self._record_format(packed)
            assert set(store) == {"actions", "rewards", "done", "truncated"} and obs is not None and obs.mode == "raw" \
                and obs.dtype == torch.uint8, "packed=True carries the full trajectory and the raw uint8 observation"
            rec = torch.empty(T, B, lay.record_bytes, dtype=torch.uint8, device=self.device)
            out["record"] = rec
            if not lay.planar and not lay.flags_packed:  # whole records: the fields are views that every launch refreshes
                out.update(self.record_fields(rec, packed))
            if replay_feed:  # (whole records only: the 1v1 kernels) the true terminal states, for susnet_ring_append reading the records
                assert not lay.planar, "replay_feed with packed records: handles that store whole records (the 1v1 kernels)"
                out["term_obs"] = torch.zeros(T, B, self.flattened_state_size, dtype=torch.uint8, device=self.device)
            # (planar records -- the multi-agent kernels: the fields are gathered after a launch, `record_fields(bufs["record"])`;
            # `rollout()` does it)
            return out
        if "actions" in store:
            out["actions"] = torch.empty(T, B, A, dtype=torch.uint8, device=self.device)
        if "rewards" in store:
            out["rewards"] = torch.empty(T, B, A, dtype=torch.float32, device=self.device)
        if "done" in store:
            out["done"] = torch.empty(T, B, dtype=torch.bool, device=self.device)
        if "truncated" in store:
            out["truncated"] = torch.empty(T, B, dtype=torch.bool, device=self.device)
        if obs is not None and obs.mode is not None:
            spec, o1, o2 = self._make_obs(obs, T)
            out["_obs_spec"] = spec
            out["obs"] = o1 if T > 1 else o1.unsqueeze(0)
            if o2 is not None:
                out["obs_non_spatial"] = o2 if T > 1 else o2.unsqueeze(0)
        if replay_feed:
            assert set(store) == {"actions", "rewards", "done", "truncated"} and obs is not None and obs.mode == "raw" \
                and obs.dtype == torch.uint8, "replay_feed=True goes with the full trajectory and the raw uint8 observation"
            out["term_obs"] = torch.zeros(T, B, self.flattened_state_size, dtype=torch.uint8, device=self.device)
            out["roles"] = torch.zeros(T, B, dtype=torch.int16, device=self.device)
        return out

    def rollout_into(self, n_ticks: int, bufs) -> None:
        """One fused launch of ``n_ticks`` (<= the buffers' capacity) ticks; asynchronous on the current stream."""
        # (numpy / tape handles: the full trajectory with the raw uint8 observation only -- susnet_rollout refuses anything else)
        assert 1 <= n_ticks <= bufs["n_ticks"]
        io = bufs.get("_io")  # the argument block is built once per buffer set (this call is on the launch-bound path)
        if io is None:
            io = L.RolloutIO()
            if "record" in bufs:
                io.record = bufs["record"].data_ptr()
                io.record_format = bufs.get("_record_format", L.RECORD_DEFAULT)
                if "term_obs" in bufs:
                    io.term_obs = bufs["term_obs"].data_ptr()
            else:
                for name in ("actions", "rewards", "done", "truncated"):
                    if name in bufs:
                        setattr(io, name, bufs[name].data_ptr())
                if "_obs_spec" in bufs:
                    io.obs = C.pointer(bufs["_obs_spec"])
                if "term_obs" in bufs:
                    io.term_obs, io.roles = bufs["term_obs"].data_ptr(), bufs["roles"].data_ptr()
            bufs["_io"] = io
        io.n_ticks = int(n_ticks)
        with self._on_device():
            L.check(self.lib.susnet_rollout(self._h, C.byref(io), self._stream()))

    def rollout(self, n_ticks: int, store=("actions", "rewards", "done", "truncated"), obs: Optional[ObsConfig] = None,
                packed: bool = False):
        """Fused random rollout (ReplayBuffer.populate's loop, reference src/replay_memory.py:96-143, without
        the buffer): ``n_ticks`` x {sample_actions; step; reset on done|truncated} in ONE launch.
        Returns a dict of trajectory tensors with a leading tick dimension."""
        bufs = self.alloc_rollout(n_ticks, store, obs, packed=packed)
        self.rollout_into(n_ticks, bufs)
        if packed and "actions" not in bufs:  # planar records / shared flag bytes: gather the fields (copies)
            bufs.update(self.record_fields(bufs["record"], packed))
        return bufs

    def observe(self, obs: Optional[ObsConfig] = None):
        """Observation of the CURRENT state (same writers the step kernel fuses)."""
        oc = obs or self.obs_config
        spec, o1, o2 = self._make_obs(oc, 1)
        with torch.cuda.device(self.device):
            L.check(self.lib.susnet_observe(self._h, C.byref(spec), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        return (o1, o2) if o2 is not None else o1

    _ROW_DTYPES = {torch.uint8: L.U8, torch.int32: L.I32, torch.int64: L.I64, torch.float32: L.F32, torch.float64: L.F64}

    def featurize(self, states: torch.Tensor, obs: ObsConfig):
        """Observation of caller-supplied FLATTENED states ``[..., S]`` (``flatten_state`` order, S =
        ``flattened_state_size``) instead of the env's own state: the reference's
        ``SequenceStateFeaturizer.fit(state_sequence[B, T, S])`` (src/features/model_ready.py:41-57) for a window
        or a replay batch.  Returns tensors with the same leading dimensions as ``states``."""
        assert obs.mode in ("flat", "planes", "persp"), "featurize() produces the flat / planes / perspective feature layouts"
        assert states.shape[-1] == self.flattened_state_size, (
            f"expected rows of {self.flattened_state_size} values, got {states.shape[-1]}")
        assert states.dtype in self._ROW_DTYPES, f"unsupported state dtype {states.dtype}"
        lead = tuple(states.shape[:-1])
        rows = states.to(self.device).reshape(-1, states.shape[-1]).contiguous()
        n = rows.shape[0]
        spec, o1, o2 = self._make_obs(obs, 1, rows=n)
        with torch.cuda.device(self.device):
            L.check(self.lib.susnet_featurize(self._h, rows.data_ptr(), self._ROW_DTYPES[rows.dtype], n, C.byref(spec), self._stream()))
            if self.check_errors:
                self.poll_errors()
        o1 = o1.reshape(*lead, *o1.shape[1:])
        if o2 is not None:
            return o1, o2.reshape(*lead, *o2.shape[1:])
        return o1

    def poll_errors(self):
        bits = C.c_uint32(0)
        rc = self.lib.susnet_poll_errors(self._h, C.byref(bits), self._stream())
        if rc == L.E_ACTION_ASSERT:
            raise AssertionError("Invalid action(s): some action >= action_space.n")  # base.py:360-362
        if rc == L.E_ACTION_INDEX:
            raise IndexError("list index out of range")  # base.py:379-382
        if rc == L.E_ROW:
            raise IndexError(self.lib.susnet_last_error().decode())  # the reference's planes would index out of bounds
        L.check(rc)

    def set_state(self, *, agent_positions=None, alive_agents=None, imposter_mask=None, job_positions=None,
                  completed_jobs=None, used_tag_actions=None, tag_counts=None, tag_reset_timer=None, t=None,
                  metrics=None, rng_cursor=None, episode_index=None):
        """Direct state assignment (what reference callers do with ``env.agent_positions[...] = ...``)."""
        keep = []

        def dev(x, dtype):
            tns = torch.as_tensor(np.asarray(x.cpu() if isinstance(x, torch.Tensor) else x)).to(dtype).contiguous().to(self.device)
            keep.append(tns)
            return tns.data_ptr()

        view = L.StateView()
        if agent_positions is not None:
            view.agent_positions = dev(agent_positions, torch.int32)
        if alive_agents is not None:
            view.alive_agents = dev(alive_agents, torch.uint8)
        if imposter_mask is not None:
            view.imposter_mask = dev(imposter_mask, torch.uint8)
        if job_positions is not None:
            view.job_positions = dev(job_positions, torch.int32)
        if completed_jobs is not None:
            view.completed_jobs = dev(completed_jobs, torch.uint8)
        if used_tag_actions is not None:
            view.used_tag_actions = dev(used_tag_actions, torch.uint8)
        if tag_counts is not None:
            view.tag_counts = dev(tag_counts, torch.int32)
        if tag_reset_timer is not None:
            view.tag_reset_timer = dev(tag_reset_timer, torch.int32)
        if t is not None:
            view.t = dev(t, torch.int32)
        if metrics is not None:
            view.metrics = dev(metrics, torch.int64)
        if rng_cursor is not None:
            view.rng_cursor = dev(rng_cursor, torch.int64)
        if episode_index is not None:
            view.episode_index = dev(episode_index, torch.int32)
        with torch.cuda.device(self.device):
            L.check(self.lib.susnet_import_state(self._h, C.byref(view), self._stream()))
            if self.export_state:
                self._export(full=True)
            torch.cuda.current_stream(self.device).synchronize()

    @property
    def tick(self) -> int:
        """Steps this handle has taken (index of the Philox action stream; see susnet_tick)."""
        v = C.c_uint64(0)
        with self._on_device():
            L.check(self.lib.susnet_tick(self._h, None, C.byref(v), self._stream()))
        return int(v.value)

    @tick.setter
    def tick(self, value: int):
        v = C.c_uint64(int(value))
        with self._on_device():
            L.check(self.lib.susnet_tick(self._h, C.byref(v), None, self._stream()))

    def device_tick(self, enable: bool = True) -> None:
        """Keep the step counter of the action stream in device memory (``susnet_device_tick``): launches then carry no
        per-call value and a captured hipGraph of them can be replayed.  ``env.tick`` keeps working (it synchronises)."""
        with self._on_device():
            L.check(self.lib.susnet_device_tick(self._h, int(bool(enable)), self._stream()))

    def capture_random_step(self, n_ticks: int = 1) -> "torch.cuda.CUDAGraph":
        """hipGraph of ``n_ticks`` drop-in ticks ``a = env.sample_actions(); env.step(a)`` (two kernel nodes each): the
        launch-bound inner loop of a random-policy driver, replayable with ``graph.replay()``.  Outputs land in the env's persistent
        tensors (``sample_actions()`` buffer, ``step()``'s rewards / done / truncated / fused observation).  Needs
        ``check_errors=False, export_state=False`` (both would synchronise inside the capture)."""
        assert self.rng_kind == "philox", "graph replay needs the counter-based production stream"
        assert not self.check_errors and not self.export_state, "construct the env with check_errors=False, export_state=False"
        self.device_tick(True)
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up on a side stream, as torch's capture rules ask (these two ticks count)
            for _ in range(2):
                self.step(self.sample_actions())
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(n_ticks):
                self.step(self.sample_actions())
        return graph

    def graph_nodes_per_tick(self) -> int:
        """Kernel nodes one captured drop-in tick holds (sample_actions + step; the step kernel advances the device-resident
        step counter itself)."""
        return 2

    def rng_cursor(self) -> torch.Tensor:
        cur = torch.zeros(self.batch, dtype=torch.int64, device=self.device)
        view = L.StateView()
        view.rng_cursor = cur.data_ptr()
        with torch.cuda.device(self.device):
            L.check(self.lib.susnet_export_state(self._h, C.byref(view), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        return cur

    def episode_index(self) -> torch.Tensor:
        """int64[B]: resets drawn so far per env = the index of its next reset in the RESET stream (production protocol: a reset's
        draws are a function of (seed, global env id, this index) alone)."""
        ep = torch.zeros(self.batch, dtype=torch.int32, device=self.device)
        view = L.StateView()
        view.episode_index = ep.data_ptr()
        with torch.cuda.device(self.device):
            L.check(self.lib.susnet_export_state(self._h, C.byref(view), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        return ep.to(torch.int64) & 0xFFFFFFFF

    def lifetime_totals(self) -> torch.Tensor:
        """int64[12] sums over this device's envs of the per-env episode accumulators (device tensor):
        the vector the multi-GPU host all-gathers (see dist.py)."""
        with torch.cuda.device(self.device):
            L.check(self.lib.susnet_reduce_lifetime(self._h, self._life_sum.data_ptr(), self._stream()))
        return self._life_sum

    def compute_action(self, agent_idx, action_idx, env_idx: int = 0):  # base.py:581-582
        return str(self.agent_action_map[agent_idx, env_idx][action_idx])


class BatchedImposterTrainingGround(BatchedFourRoomEnv):
    """reference src/environment/pred_prey.py:20-99: one imposter, no FIX/SABOTAGE, fixed action order,
    dead_penalty 0, own win rule; allows 1v1 (75-76)."""

    VARIANT = L.VARIANT_ITG
    crew_actions = CREW_ACTIONS_SIMPLE
    imposter_actions = IMPOSTER_ACTIONS_SIMPLE

    def __init__(self, n_crew, n_jobs, time_step_reward, kill_reward, sabotage_reward, end_of_game_reward,
                 random_state=None, debug=False, shuffle_imposter_index=False, include_walls: bool = True, **batched):
        super().__init__(
            n_imposters=1, n_crew=n_crew, n_jobs=n_jobs, time_step_reward=time_step_reward, kill_reward=kill_reward,
            sabotage_reward=sabotage_reward, debug=debug, dead_penalty=0, game_end_reward=end_of_game_reward,
            random_state=random_state, is_action_order_random=False, shuffle_imposter_index=shuffle_imposter_index,
            include_walls=include_walls, **batched)

    def _validate_init_args(self, n_imposters, n_crew, n_jobs):  # pred_prey.py:75-76
        assert n_crew > 0, f"Must have at least one crew member. Got {n_crew}."


class BatchedFourRoomEnvWithTagging(BatchedFourRoomEnv):
    """reference src/environment/tagging.py:9-249 (per-agent one-shot votes, tally every tag_reset_interval)."""

    VARIANT = L.VARIANT_TAGGING

    def __init__(self, *args, tag_reset_interval: int = 50, vote_reward: int = 3, **kwargs):
        self.tag_reset_interval, self.vote_reward = tag_reset_interval, vote_reward
        super().__init__(*args, _tag_reset_interval=tag_reset_interval, _vote_reward=vote_reward, **kwargs)
        self.n_imposter_actions += self.n_agents - 1  # tagging.py:35-36
        self.n_crew_actions += self.n_agents - 1
        # tagging.py:15-28 (this order does not match the returned tuple -- reproduced as is)
        self.state_fields = {f: i for i, f in enumerate(
            [StateFields.AGENT_POSITIONS, StateFields.JOB_POSITIONS, StateFields.JOB_STATUS, StateFields.ALIVE_AGENTS,
             StateFields.USED_TAGS, StateFields.TAG_COUNTS, StateFields.TAG_RESET_COUNT])}

    @property
    def tag_reset_timer(self):
        return self._timer

    def _state_tuple(self):  # tagging.py:94-99, 220-230
        return (self.agent_positions, self.alive_agents, self.job_positions, self.completed_jobs,
                self.used_tag_actions, self.tag_counts, self.tag_reset_interval - self._timer)

    def reset(self, seed=None, mask=None, **kwargs):
        state, _ = super().reset(seed=seed, mask=mask, **kwargs)
        return state, {}  # tagging.py:101

    def compute_action(self, agent_idx, action_idx, env_idx: int = 0):  # tagging.py:243-249
        if action_idx < len(Action):
            return str(Action(action_idx))
        players = [p for p in range(self.n_agents) if p != agent_idx]
        return f"Vote Player {players[action_idx - len(Action)]}"


class _MetricsView:
    """``env.metrics`` of the reference (EnvMetricHandler, src/metrics.py:35-64), batched: values are [B] tensors."""

    def __init__(self, env):
        self._env = env
        self._views = None

    def get_metrics(self) -> Dict[SusMetrics, torch.Tensor]:
        # column views of the persistent [B, 13] export buffer: built once, they alias it like the reference's
        # state arrays alias the env's (refreshed in place by every step)
        if self._views is None:
            m = self._env._metrics
            self._views = {metric: m[:, i] for i, metric in enumerate(SusMetrics)}
        return self._views

    @property
    def metrics(self):
        return self.get_metrics()

    def __repr__(self):
        return repr({k.value: v.tolist() for k, v in self.get_metrics().items()})


class _ActionMapView:
    """``env.agent_action_map[agent]`` (base.py:306-312; tagging.py:68-75) for environment ``env_idx``."""

    def __init__(self, env):
        self._env = env

    def __getitem__(self, key):
        agent, env_idx = key if isinstance(key, tuple) else (key, 0)
        e = self._env
        is_imp = bool(e.imposter_mask[env_idx, agent])
        acts = list(e.imposter_actions if is_imp else e.crew_actions)
        if e.VARIANT == L.VARIANT_TAGGING:
            acts += [p for p in range(e.n_agents) if p != agent]
        return acts

    def __iter__(self):
        return iter(range(self._env.n_agents))

    def __len__(self):
        return self._env.n_agents
