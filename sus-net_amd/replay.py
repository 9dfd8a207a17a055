"""Device-side sequence window + replay ring: the immediate consumer of the
environment path, kept on the GPU so that batched rollouts become training batches without host copies.

Mirrors the reference's ``ReplayBuffer`` (src/replay_memory.py:11-94): same tensors and shapes
(``states [max, T, S]`` float32, ``actions [max, A]`` int64, ``rewards [max, A]`` float32,
``next_states [max, T, S]``, ``dones [max, 1]`` bool, ``imposters [max, n_imp]`` int16), same ring arithmetic,
same ``sample`` (uniform with replacement over the filled part) -- but ``add_batch`` writes B transitions at a
time and everything lives on the env's device.  ``populate`` is the batched form of ``ReplayBuffer.populate``
(src/replay_memory.py:96-143) with the ``np.roll`` sequence window of train.py:388-389: every env keeps the
window of its last T flattened states (O1 layout, written by the step kernel's fused raw observation); a fresh
episode's window is filled with its first state, as the reference does after ``reset``.

Two ways to fill the ring:
* ``populate_fused``: the native path -- fused rollout launches (``susnet_rollout`` with its replay feed) followed by ONE
  ``susnet_ring_append`` launch each, which writes all six reference tensors for T x B transitions from the trajectory with
  exactly ``ReplayBuffer.populate``'s window / terminal-state semantics (pinned against the reference's own buffers,
  tests/golden/replay_*.npz).
* ``populate`` / ``add_batch``: per-tick torch glue over the step API (kept for callers that step the env themselves).

One documented difference: ``imposters`` holds each episode's imposter indices in ASCENDING order (the kernels carry the
roles as a bitmask).  The reference stores ``env.imposter_idxs`` in numpy's draw order (``np.random.choice``, base.py:274),
so with ``n_imposters >= 2`` and ``shuffle_imposter_index`` a row can read ``[5, 2]`` there and ``[2, 5]`` here -- the same
set; every consumer in the reference (train.py:398-416) uses the row as a set (``agent_idx in imposters``).  With one
imposter, and with the shuffle off, the rows are identical.
"""
from __future__ import annotations

from collections import namedtuple

import torch

Batch = namedtuple("Batch", ("states", "actions", "rewards", "next_states", "imposters", "dones"))  # replay_memory.py:6-8


class DeviceReplayBuffer:
    # field -> (per-row shape builder, dtype): the reference's tensor zoo (replay_memory.py:33-44) as a table
    _FIELDS = {
        "states": (lambda o: (o.trajectory_size, o.state_size), torch.float32),
        "actions": (lambda o: (o.n_agents,), torch.long),
        "rewards": (lambda o: (o.n_agents,), torch.float32),
        "next_states": (lambda o: (o.trajectory_size, o.state_size), torch.float32),
        "dones": (lambda o: (1,), torch.bool),
        "imposters": (lambda o: (o.n_imposters,), torch.int16),
    }

    def __init__(self, max_size: int, state_size: int, trajectory_size: int, n_agents: int, n_imposters: int, device="cuda"):
        for name, value in (("Replay buffer size", max_size), ("Trajectory size", trajectory_size), ("State size", state_size),
                            ("Number of agents", n_agents)):
            assert value > 0, f"{name} must be positive"
        self.max_size, self.trajectory_size, self.state_size = max_size, trajectory_size, state_size
        self.n_agents, self.n_imposters = n_agents, n_imposters
        dev = torch.device(device)
        for field, (shape, dtype) in self._FIELDS.items():
            setattr(self, field, torch.empty((max_size, *shape(self)), dtype=dtype, device=dev))
        self.idx = self.size = 0  # ring cursor and fill level

    def add_batch(self, state, action, reward, next_state, done, imposters) -> None:
        """N transitions at once; equivalent to N reference ``add`` calls in row order (replay_memory.py:50-73)."""
        n = state.shape[0]
        if n > self.max_size:  # only the last max_size rows would survive N sequential adds
            cut = n - self.max_size
            state, action, reward, next_state, done, imposters = (x[cut:] for x in (state, action, reward, next_state, done, imposters))
            self.idx = (self.idx + cut) % self.max_size
            n = self.max_size
        pos = (self.idx + torch.arange(n, device=self.states.device)) % self.max_size
        self.states[pos] = state.to(self.states.dtype)
        self.actions[pos] = action.to(torch.long)
        self.rewards[pos] = reward.to(self.rewards.dtype)
        self.next_states[pos] = next_state.to(self.next_states.dtype)
        self.dones[pos] = done.reshape(n, 1).to(torch.bool)
        self.imposters[pos] = imposters.to(torch.int16)
        self.idx = (self.idx + n) % self.max_size
        self.size = min(self.size + n, self.max_size)

    def sample(self, batch_size) -> Batch:  # replay_memory.py:75-94
        assert self.size > 0, "Replay buffer is empty, can't sample"
        i = torch.randint(0, self.size, (batch_size,), device=self.states.device)
        return Batch(states=self.states[i], actions=self.actions[i], rewards=self.rewards[i], next_states=self.next_states[i],
                     imposters=self.imposters[i], dones=self.dones[i])

    @torch.no_grad()
    def populate_fused(self, env, num_steps: int, ticks_per_launch: int = 256, packed=False) -> int:
        """``ReplayBuffer.populate`` (src/replay_memory.py:96-143) for B environments in lockstep, on the device:
        ``env.reset()``, then ``num_steps`` ticks of the fused random rollout, appended to the ring by ``susnet_ring_append``.
        Row order = tick-major, env-minor (what B sequential ``add`` calls per tick produce); with ``batch=1`` and
        ``rng='numpy'`` the ring equals the reference's for the same numpy seed.  ``env`` must auto-reset.  ``packed`` (True / "compact";
        handles that store whole records: the 1v1 kernels): the rollout writes packed records and ``susnet_ring_append`` reads them in
        place -- no separate trajectory tensors."""
        import ctypes as C

        from . import _lib as L
        from .env import ObsConfig

        assert env.auto_reset, "populate_fused needs an auto-resetting env (episodes restart inside the launch)"
        assert env.flattened_state_size == self.state_size and env.n_agents == self.n_agents and env.n_imposters == self.n_imposters
        T, B, S = self.trajectory_size, env.batch, self.state_size
        raw8 = ObsConfig("raw", dtype=torch.uint8)
        env.reset()
        first = env.observe(raw8)
        window = first.unsqueeze(1).repeat(1, T, 1).contiguous()  # replay_memory.py:108-113: the first state T times
        n_launch = max(1, min(int(ticks_per_launch), int(num_steps)))
        bufs = env.alloc_rollout(n_launch, obs=raw8, replay_feed=True, packed=packed)
        io = L.RingIO()
        io.trajectory_size = T
        if packed:
            io.record, io.record_format, io.term_obs = bufs["record"].data_ptr(), bufs["_record_format"], bufs["term_obs"].data_ptr()
        else:
            io.actions, io.rewards = bufs["actions"].data_ptr(), bufs["rewards"].data_ptr()
            io.done, io.truncated = bufs["done"].data_ptr(), bufs["truncated"].data_ptr()
            io.obs, io.term_obs, io.roles = bufs["obs"].data_ptr(), bufs["term_obs"].data_ptr(), bufs["roles"].data_ptr()
        io.window = window.data_ptr()
        io.max_size = self.max_size
        io.states, io.next_states = self.states.data_ptr(), self.next_states.data_ptr()
        io.ring_actions, io.ring_rewards = self.actions.data_ptr(), self.rewards.data_ptr()
        io.ring_dones, io.ring_imposters = self.dones.data_ptr(), self.imposters.data_ptr()
        done_ticks = 0
        while done_ticks < num_steps:
            n = min(n_launch, num_steps - done_ticks)
            env.rollout_into(n, bufs)
            io.n_ticks, io.idx = n, self.idx
            with torch.cuda.device(env.device):
                L.check(env.lib.susnet_ring_append(env._h, C.byref(io), env._stream()))
            self.idx = (self.idx + n * B) % self.max_size
            self.size = min(self.size + n * B, self.max_size)
            done_ticks += n
        torch.cuda.current_stream(env.device).synchronize()  # (the trajectory buffers die with this call)
        return num_steps * B

    def _ring_io(self, env, feed, window):
        import ctypes as C

        from . import _lib as L

        io = L.RingIO()
        io.trajectory_size = self.trajectory_size
        io.actions, io.rewards = feed["actions"].data_ptr(), feed["rewards"].data_ptr()
        io.done, io.truncated = feed["done"].data_ptr(), feed["truncated"].data_ptr()
        io.obs, io.term_obs, io.roles = feed["obs"].data_ptr(), feed["term_obs"].data_ptr(), feed["roles"].data_ptr()
        io.window = window.data_ptr()
        io.max_size = self.max_size
        io.states, io.next_states = self.states.data_ptr(), self.next_states.data_ptr()
        io.ring_actions, io.ring_rewards = self.actions.data_ptr(), self.rewards.data_ptr()
        io.ring_dones, io.ring_imposters = self.dones.data_ptr(), self.imposters.data_ptr()
        return io

    @torch.no_grad()
    def collect(self, env, policy, num_steps: int, epsilon: float = 0.0, mask_dead: bool = True, ticks_per_append: int = 64,
                one_launch_per_block: bool = True) -> int:
        """The trainer's collection loop (train.py:345-399) for B environments in lockstep, on the device: per tick the teams act by
        their Q-networks on the current state -- epsilon-greedy, dead agents get index 0 (train.py:351-381) --, the env steps, and the
        transition ``(window, actions, rewards, next window, done, imposters)`` goes to the ring with ``ReplayBuffer.add``'s semantics
        (replay_memory.py:50-73; windows as train.py:318-322, 388-389, 441-445).  ``policy``: a ``PolicyRollout`` over ``env`` whose
        models are reference MLPs the Q-network kernel serves (``policy.fused_imposter``; the crew by ``policy.fused_crew``, or random
        when it has no model).  Per tick ONE kernel where the env serves the whole tick (``susnet_qnet_policy_step``), else the network
        launch(es) + ``susnet_policy_step``; every ``ticks_per_append`` ticks ONE ``susnet_ring_append``.  ``one_launch_per_block``: the block's
        ticks in ONE launch where the env serves it (``susnet_qnet_policy_rollout``: the weights are fixed within a block anyway).  The sequence window carries
        over between calls (``reset_collection`` after an ``env.reset()``).  Row order = tick-major, env-minor; with ``batch=1``,
        ``rng='numpy'`` and two networks the ring equals the reference's for the same numpy seed and weights
        (tests/golden/collect_*.npz).  Returns the number of transitions added."""
        import ctypes as C

        from . import _lib as L
        from .env import ObsConfig

        assert policy.env is env and policy.fused_imposter is not None, "collect: the imposters' model must be a reference MLP on a compiled-in feature layout"
        assert policy.crew_model is None or policy.fused_crew is not None, "collect: the crew's model must be a reference MLP on a compiled-in feature layout (or None: random crew)"
        assert env.flattened_state_size == self.state_size and env.n_agents == self.n_agents and env.n_imposters == self.n_imposters
        T, B = self.trajectory_size, env.batch
        # the carried window and the feed block belong to ONE env (identity, device) between two of its resets: another env of the same
        # batch, or the same env after reset(), starts from its own current state
        owner = (id(env), str(env.device), getattr(env, "reset_generation", 0))
        if getattr(self, "_collect_owner", None) != owner:
            self._collect_window = None
            if getattr(self, "_collect_owner", (None, None, None))[:2] != owner[:2]:
                self._collect_feed = None
            self._collect_owner = owner
        if getattr(self, "_collect_window", None) is None:  # train.py:318-322: the current state T times
            first = env.observe(ObsConfig("raw", dtype=torch.uint8))
            self._collect_window = first.unsqueeze(1).repeat(1, T, 1).contiguous()
        n_block = max(1, min(int(ticks_per_append), int(num_steps)))
        feed = getattr(self, "_collect_feed", None)
        if feed is None or feed["n_ticks"] != n_block or feed["actions"].shape[1] != B:
            feed = self._collect_feed = env.alloc_feed(n_block)
        io = self._ring_io(env, feed, self._collect_window)
        # one launch per BLOCK where the env serves the whole tick in one kernel and the block's observation slots are 16-byte aligned
        fused_block = (one_launch_per_block and env.supports_qnet_policy_step(policy.fused_imposter, policy.fused_crew, epsilon)
                       and (n_block == 1 or (B * self.state_size) % 16 == 0))
        done_ticks = 0
        while done_ticks < num_steps:
            n = min(n_block, num_steps - done_ticks)
            policy.refresh_weights(force=False)  # (the optimizer may have stepped since the last block)
            if fused_block:  # the whole block in ONE launch (susnet_qnet_policy_rollout)
                env.policy_rollout_into(feed, n, policy.fused_imposter, epsilon=epsilon, mask_dead=mask_dead, net_crew=policy.fused_crew)
            else:
                for t in range(n):
                    env.policy_tick_into(feed, t, net_imposter=policy.fused_imposter, net_crew=policy.fused_crew, epsilon=epsilon, mask_dead=mask_dead)
            io.n_ticks, io.idx = n, self.idx
            with torch.cuda.device(env.device):
                L.check(env.lib.susnet_ring_append(env._h, C.byref(io), env._stream()))
            self.idx = (self.idx + n * B) % self.max_size
            self.size = min(self.size + n * B, self.max_size)
            done_ticks += n
        return num_steps * B

    def reset_collection(self) -> None:
        """Forget the carried sequence window: the next ``collect`` starts from the current state.  (``collect`` does this by itself when the
        env it is given is another one, or has been ``reset()`` since the last call.)"""
        self._collect_window = None

    @torch.no_grad()
    def populate(self, env, num_steps: int) -> int:
        """Random-policy rollout into the ring: ``num_steps`` lockstep ticks of ``env`` (B transitions each).
        ``env`` must fuse the float32 raw observation (``obs=ObsConfig('raw')``).

        * ``auto_reset=False`` env: EXACT reference semantics -- the stored ``next_states`` window ends with the true
          post-step state (also for terminal and truncated transitions, replay_memory.py:120-136); ended envs are then
          reset with a masked ``env.reset`` (one more small launch per tick).
        * ``auto_reset=True`` env: one launch per tick; for a transition that ENDS an episode the last row of
          ``next_states`` is already the new episode's first state.  ``done`` is stored, so a learner never bootstraps
          across a terminal boundary; a truncated transition (done = False) does bootstrap from the fresh state.
        """
        assert env.obs_config.mode == "raw", "populate needs the fused raw observation"
        assert env.obs.shape[-1] == self.state_size
        T = self.trajectory_size
        env.reset()
        window = env.obs.unsqueeze(1).repeat(1, T, 1)  # replay_memory.py:112-113: the first state T times
        added = 0
        for _ in range(num_steps):
            if not env.export_state:
                env.refresh_roles()
            imposters = env.imposter_idxs  # roles of the acting episode (replay_memory.py:117)
            a = env.sample_actions().clone()
            _, rew, done, trunc, _ = env.step(a)
            ended = done | trunc
            nxt = torch.roll(window, shifts=-1, dims=1)  # replay_memory.py:122-127 / train.py:388-389
            nxt[:, -1] = env.obs
            self.add_batch(window, a, rew, nxt, done, imposters)
            if not env.auto_reset:
                env.reset(mask=ended)  # fuses the fresh states of the ended envs into env.obs
            window = torch.where(ended.view(-1, 1, 1), env.obs.unsqueeze(1).expand(-1, T, -1), nxt)
            added += env.batch
        return added
