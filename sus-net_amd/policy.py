"""Policy-in-the-loop rollout (BASELINE.json config 5).

The loop around the HIP environment -- fused flat observation -> model -> argmax -> `step`, greedy as in the reference's
`run_game` (src/visualize.py:547-582) -- with every tensor staying on the device.  The networks are the reference's architectures
as torch modules (so that its checkpoints load); a reference ``MLP`` on one of the compiled-in feature layouts does not RUN through
torch, though: ``PolicyRollout`` hands its weights to the library once (``pack_mlp`` -> ``susnet_qnet_pack``) and a tick is then
one kernel (``susnet_qnet_policy_step``: network on the f32-input MFMA, argmax, the crew's draws, the env step), or two
(``susnet_qnet_forward`` + ``susnet_policy_step``) when the crew has a network too.  Everything else (``SpatialDQN``, other layer
stacks) runs through stock PyTorch-ROCm and hands its Q rows to ``susnet_policy_step`` / ``susnet_policy_actions``.

* ``MLP`` mirrors reference src/models/dqn.py:72-108 (`make_mlp` 322-329: Linear + PReLU, last activation
  dropped) INCLUDING the module names, so a reference checkpoint `{"state_dict", "config"}`
  (dqn.py:92-103) loads unchanged.  The reference ships no checkpoints (`.gitignore:2`), so benchmarks use a
  seeded random initialisation of the same architecture (layer dims of notebooks/experiment_1v1.ipynb cell 1).
* ``RandomEquiprobable`` mirrors dqn.py:111-138; in the loop a random crew is sampled by the environment's own
  `sample_actions` kernel (uniform over the role-valid indices, base.py:326-330).
* ``SpatialDQN`` mirrors dqn.py:205-314 (CNN over the plane features of every window step -> concat with the
  non-spatial features -> RNN over the window -> PReLU MLP head on the last hidden state), again with the reference's
  module names and its layer-list quirks, so reference checkpoints load; ``tests/golden/model_spatialdqn.npz`` pins it
  against the reference module's own forward pass.
* ``WindowedPolicyRollout`` is the batched form of the reference's acting loop with a state window
  (train.py:316-389, visualize.py:502-585): a ``[B, T, S]`` window of flattened states lives on the device, every
  tick it goes through ``susnet_featurize`` (Perspective / Global / Flat featurizer), each agent's view feeds the
  imposter or the crew network by the env's role mask, argmax (optionally epsilon-greedy, train.py:355-381), step,
  window roll (np.roll, train.py:388-389) or refill with the fresh first state where an episode ended.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from .env import ObsConfig


def make_mlp(layer_dims: Sequence[int]) -> nn.Sequential:  # dqn.py:322-329 with PReLU (dqn.py:79)
    layers: List[nn.Module] = []
    for idx, dim in enumerate(layer_dims[:-1]):
        layers.append(nn.Linear(in_features=dim, out_features=layer_dims[idx + 1]))
        layers.append(nn.PReLU())
    return nn.Sequential(*layers[:-1])


class MLP(nn.Module):
    def __init__(self, layer_dims):
        super().__init__()
        self.layer_dims = list(layer_dims)
        self.model = make_mlp(self.layer_dims)
        self.config = {"layer_dims": self.layer_dims}

    def forward(self, spatial_x, non_spatial_x):  # dqn.py:84-88: the spatial input is ignored
        batch_size = spatial_x.size(0)
        return self.model(non_spatial_x.reshape(batch_size, -1))

    def dump_to_checkpoint(self, filepath):  # dqn.py:90-93
        torch.save({"state_dict": self.state_dict(), "config": self.config}, filepath)

    @staticmethod
    def load_from_checkpoint(filepath, map_location=None):  # dqn.py:95-101
        checkpoint = torch.load(filepath, map_location=map_location)
        model = MLP(**checkpoint["config"])
        model.load_state_dict(checkpoint["state_dict"])
        return model

    def create_copy(self):
        new = MLP(**self.config)
        new.load_state_dict(self.state_dict())
        return new


class RandomEquiprobable(nn.Module):  # dqn.py:111-138
    def __init__(self, n_outputs: int):
        super().__init__()
        self.n_outputs = n_outputs

    def forward(self, *inputs):
        batch = inputs[0].shape[0] if inputs else 1
        dev = inputs[0].device if inputs else None
        idx = torch.randint(0, self.n_outputs, (batch,), device=dev)
        out = torch.zeros(batch, self.n_outputs, device=dev)
        out[torch.arange(batch, device=dev), idx] = 1
        return out


def calculate_cnn_output_dim(input_size, kernel_size, strides, paddings, dilations):  # src/utils.py:5-11
    size = input_size
    for stride, padding, dilation in zip(strides, paddings, dilations):
        size = (size + 2 * padding - dilation * (kernel_size[0] - 1) - 1) // stride + 1
    return size


class _CNN(nn.Module):  # dqn.py:141-181; attribute name `model` = the checkpoint key prefix "cnn.model."
    def __init__(self, n_channels, strides, paddings, kernel_size, dilations):
        super().__init__()
        # the reference repeats the LAST entry of every list once more and zips them: with len(n_channels) ==
        # len(strides) + 1 that appends one extra same-width convolution (dqn.py:155-159)
        chans = list(n_channels) + [n_channels[-1]]
        strides, paddings, dilations = (list(v) + [v[-1]] for v in (strides, paddings, dilations))
        layers: List[nn.Module] = []
        for idx, (c_in, stride, padding, dilation) in enumerate(zip(chans[:-1], strides, paddings, dilations)):
            layers += [nn.Conv2d(c_in, chans[idx + 1], kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation),
                       nn.ReLU()]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class _RNN(nn.Module):  # dqn.py:184-202
    def __init__(self, input_dim, n_layers, hidden_dim, dropout):
        super().__init__()
        self.model = nn.RNN(input_size=input_dim, hidden_size=hidden_dim, num_layers=n_layers, dropout=dropout, batch_first=True)

    def forward(self, x):
        return self.model(x)


class SpatialDQN(nn.Module):
    def __init__(self, input_image_size, non_spatial_input_size, n_channels, strides, paddings, kernel_size, dilations,
                 rnn_layers, rnn_hidden_dim, rnn_dropout, mlp_hidden_layer_dims, n_actions):
        super().__init__()
        self.config = dict(input_image_size=input_image_size, non_spatial_input_size=non_spatial_input_size,
                           n_channels=list(n_channels), strides=list(strides), paddings=list(paddings), dilations=list(dilations),
                           kernel_size=kernel_size, rnn_layers=rnn_layers, rnn_hidden_dim=rnn_hidden_dim, rnn_dropout=rnn_dropout,
                           mlp_hidden_layer_dims=list(mlp_hidden_layer_dims), n_actions=n_actions)
        self.cnn_ouput_dim = calculate_cnn_output_dim(input_image_size, kernel_size, strides, paddings, dilations)  # (sic)
        self.cnn = _CNN(n_channels, strides, paddings, kernel_size, dilations)
        self.rnn_in_dim = self.cnn_ouput_dim ** 2 * n_channels[-1] + non_spatial_input_size
        self.rnn = _RNN(self.rnn_in_dim, rnn_layers, rnn_hidden_dim, rnn_dropout)
        self.n_actions = n_actions
        self.mlp_dims = [rnn_hidden_dim] + list(mlp_hidden_layer_dims) + [n_actions]
        self.prediction_head = make_mlp(self.mlp_dims)

    def forward(self, spatial_x, non_spatial_x):  # dqn.py:275-293
        batch, steps, C, H, W = spatial_x.shape
        feats = self.cnn(spatial_x.reshape(batch * steps, C, H, W)).reshape(batch, steps, -1)
        out, _ = self.rnn(torch.cat((feats, non_spatial_x), dim=2))
        return self.prediction_head(out[:, -1, :])

    def dump_to_checkpoint(self, filepath):  # dqn.py:295-298
        torch.save({"state_dict": self.state_dict(), "config": self.config}, filepath)

    @staticmethod
    def load_from_checkpoint(filepath, map_location=None):  # dqn.py:300-306
        checkpoint = torch.load(filepath, map_location=map_location)
        model = SpatialDQN(**checkpoint["config"])
        model.load_state_dict(checkpoint["state_dict"])
        return model

    def create_copy(self):
        new = SpatialDQN(**self.config)
        new.load_state_dict(self.state_dict())
        return new


def reference_imposter_mlp(env, components: Sequence[str], seed: int = 0) -> MLP:
    """`[F, 256, 128, 64, 16, n_imposter_actions]` (notebooks/experiment_1v1.ipynb cell 1), seeded init."""
    spec, o1, _ = env._make_obs(ObsConfig("flat", list(components)), 1)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        model = MLP([o1.shape[-1], 256, 128, 64, 16, env.n_imposter_actions])
    return model.to(env.device).eval()


def reference_crew_mlp(env, components: Sequence[str], seed: int = 1) -> MLP:
    """The crew's network of the same architecture, `[F, 256, 128, 64, 16, n_crew_actions]` (train_crew, notebooks/experiment.ipynb cell 5;
    run_game's crew_model, visualize.py:547-562), seeded init."""
    spec, o1, _ = env._make_obs(ObsConfig("flat", list(components)), 1)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        model = MLP([o1.shape[-1], 256, 128, 64, 16, env.n_crew_actions])
    return model.to(env.device).eval()


def pack_mlp(env, model, components: Sequence[str], into=None):
    """``env.qnet_pack`` of a reference ``MLP`` (Linear / nn.PReLU() alternating, dqn.py:322-329), or None if ``model`` is something
    else or the library does not serve its shape on this env.  ``into``: an image of the same stack to overwrite in place."""
    if not isinstance(model, MLP):
        return None
    layers = list(model.model)
    linears, acts = layers[0::2], layers[1::2]
    if not all(isinstance(m, nn.Linear) and m.bias is not None for m in linears) or \
            not all(isinstance(m, nn.PReLU) and m.weight.numel() == 1 for m in acts) or len(acts) != len(linears) - 1:
        return None
    cpu = lambda t: t.detach().to("cpu", torch.float32).numpy()
    return env.qnet_pack(components, [cpu(m.weight) for m in linears], [cpu(m.bias) for m in linears], [float(m.weight.detach()) for m in acts], into=into)


def _weights_version(model) -> int:
    """Changes whenever a parameter of ``model`` is written in place (optimizer.step, load_state_dict, target sync) or replaced."""
    return 0 if model is None else sum(p._version + id(p) for p in model.parameters())


class PolicyRollout:
    """Greedy policy-in-the-loop stepping of a batched env.

    Per tick: the flat observation (written by the previous step / reset kernel) feeds ``imposter_model`` and
    ``crew_model``; agent *i* of env *b* takes the imposter model's argmax if it is an imposter there, else the
    crew model's (all agents see the same flat features, as in the reference's FlatFeaturizer,
    model_ready.py:356-367).  ``crew_model=None`` = uniformly random crew via the env's sample_actions kernel.
    """

    def __init__(self, env, imposter_model: nn.Module, crew_model: Optional[nn.Module] = None,
                 components: Sequence[str] = ("onehot_pos",), fused: bool = True, epsilon: float = 0.0, mask_dead: bool = False):
        assert env.obs_config.mode == "flat" and list(env.obs_config.components) == list(components), (
            "construct the env with obs=ObsConfig('flat', components) so that step() fuses the observation")
        self.env, self.imposter_model, self.crew_model = env, imposter_model, crew_model
        # epsilon-greedy acting as in the trainer (train.py:355-381; `epsilon` may be changed between ticks: the scheduler's value) and its
        # habit of giving dead agents index 0; both are applied by the kernels that choose the actions (PHILOX handles)
        self.epsilon, self.mask_dead = float(epsilon), bool(mask_dead)
        # reference MLPs on a compiled-in feature layout run as ONE kernel from the state words to the Q row (susnet_qnet_forward);
        # anything else (SpatialDQN, other layer stacks / layouts) goes through the torch module on env.obs
        self.components = list(components)
        self.fused_imposter = pack_mlp(env, imposter_model, components) if fused else None
        self.fused_crew = pack_mlp(env, crew_model, components) if fused and crew_model is not None else None
        self._packed_version = (_weights_version(imposter_model), _weights_version(crew_model))
        # ... and with a random crew on one of the compiled-in games the whole tick -- network, argmax, the crew's draws, the step -- is
        # ONE kernel (susnet_qnet_policy_step); so it is with both teams' networks (round 5: the LDS image is swapped between the two passes)
        self.one_kernel_tick = (self.fused_imposter is not None and (crew_model is None or self.fused_crew is not None) and
                                env.supports_qnet_policy_step(self.fused_imposter, self.fused_crew, self.epsilon))
        B = env.batch
        self._spatial = torch.zeros(B, 1, 1, device=env.device)  # FlatFeaturizer's dummy spatial input
        self._actions = torch.zeros(B, env.n_agents, dtype=torch.int64, device=env.device)

    def refresh_weights(self, force: bool = True) -> bool:
        """Re-pack the models' CURRENT weights into the device images the fused kernels read (in place: a captured graph keeps
        reading the same buffers).  The acting loop of the trainer changes the weights between ticks (optimizer.step, target sync:
        train.py:402-416); ``tick()`` / ``act()`` / ``q_rows()`` call this with ``force=False`` -- a version check of the parameters --
        so eager loops follow the modules by themselves, but a REPLAYED graph runs no host code: call ``refresh_weights()`` before
        ``graph.replay()`` after changing weights.  Returns whether anything was re-packed."""
        ver = (_weights_version(self.imposter_model), _weights_version(self.crew_model))
        if not force and ver == self._packed_version:
            return False
        if self.fused_imposter is not None:
            pack_mlp(self.env, self.imposter_model, self.components, into=self.fused_imposter)
        if self.fused_crew is not None:
            pack_mlp(self.env, self.crew_model, self.components, into=self.fused_crew)
        self._packed_version = ver
        return self.fused_imposter is not None or self.fused_crew is not None

    @torch.no_grad()
    def q_rows(self):
        """The teams' Q rows on the current observation: ``(q_imposter [B, n_imposter_actions], q_crew or None)``."""
        env = self.env
        self.refresh_weights(force=False)
        feats = env.obs  # [B, F] float32, refreshed by reset()/step()
        q_imp = env.qnet_forward(self.fused_imposter) if self.fused_imposter is not None else self.imposter_model(self._spatial, feats)
        q_crew = None
        if self.crew_model is not None:
            q_crew = env.qnet_forward(self.fused_crew) if self.fused_crew is not None else self.crew_model(self._spatial, feats)
        return q_imp.contiguous(), (q_crew.contiguous() if q_crew is not None else None)

    @torch.no_grad()
    def tick(self):
        """One tick of the acting loop (visualize.py:547-582): Q rows, greedy actions, env.step.  Returns ``(actions, rewards, done,
        truncated)``.  ONE launch where the env serves it (``susnet_qnet_policy_step``: a reference MLP for the imposters, a random crew, a
        compiled-in game); else two (the network, then ``susnet_policy_step``: argmax, the crew's draws and the step in one kernel); else
        ``act()`` + ``env.step``."""
        env = self.env
        if self.one_kernel_tick:
            self.refresh_weights(force=False)
            _, rew, done, trunc, _, a = env.qnet_policy_step(self.fused_imposter, actions_out=self._actions, epsilon=self.epsilon, mask_dead=self.mask_dead,
                                                             net_crew=self.fused_crew)
            return a, rew, done, trunc
        q_imp, q_crew = self.q_rows()
        fits = max(env.n_imposter_actions, env.n_crew_actions) <= 16 and (q_crew is not None or env.rng_kind == "philox")
        if fits:
            _, rew, done, trunc, _, a = env.policy_step(q_imp, q_crew, actions_out=self._actions, epsilon=self.epsilon, mask_dead=self.mask_dead)
            return a, rew, done, trunc
        a = self.act()
        _, rew, done, trunc, _ = env.step(a)
        return a, rew, done, trunc

    @torch.no_grad()
    def act(self) -> torch.Tensor:
        """One tick's actions: the network forward(s) through stock PyTorch-ROCm, then ONE kernel (``susnet_policy_actions``) that
        reads the episode's roles from the env's state, takes each team's argmax and -- without a crew network -- the crew's
        draws from the action stream.  (The eager form of the same thing -- role export, argmax, sample_actions, dtype copy,
        where -- was five launches and a tenth of the tick's GPU time.)"""
        env = self.env
        q_imp, q_crew = self.q_rows()
        if env.rng_kind != "philox" and q_crew is None:  # numpy tapes: the crew's draws come from the env's own words (greedy only)
            assert not self.epsilon and not self.mask_dead, "epsilon-greedy / mask_dead act through the production stream: rng='philox'"
            if not env.export_state:
                env.refresh_roles()
            torch.where(env.imposter_mask, q_imp.argmax(dim=1).unsqueeze(1), env.sample_actions().to(torch.int64), out=self._actions)
            return self._actions
        return env.policy_actions(q_imp, q_crew, out=self._actions, epsilon=self.epsilon, mask_dead=self.mask_dead)

    @torch.no_grad()
    def run(self, n_steps: int, record: bool = False, block_ticks: int = 64) -> Dict[str, torch.Tensor]:
        """``n_steps`` ticks of the acting loop with the networks as they are (``run_game``, visualize.py:547-582).  Where the env serves the
        whole tick as one kernel and nothing is recorded, ``block_ticks`` ticks go into ONE launch (``susnet_qnet_policy_rollout``; 0: one
        launch per tick)."""
        env = self.env
        if self.one_kernel_tick and not record and block_ticks > 0 and env.auto_reset:
            self.refresh_weights(force=False)
            left = int(n_steps)
            while left > 0:
                n = min(left, int(block_ticks))
                env.policy_block(n, self.fused_imposter, epsilon=self.epsilon, mask_dead=self.mask_dead, net_crew=self.fused_crew)
                left -= n
            return {}
        out: Dict[str, List[torch.Tensor]] = {"actions": [], "rewards": [], "done": [], "truncated": []}
        for _ in range(n_steps):
            a, rew, done, trunc = self.tick()
            if record:
                out["actions"].append(a.clone())
                out["rewards"].append(rew.clone())
                out["done"].append(done.clone())
                out["truncated"].append(trunc.clone())
        return {k: torch.stack(v) for k, v in out.items()} if record else {}


    @torch.no_grad()
    def capture(self, n_ticks: int = 8, record: bool = False):
        """The policy tick -- {roles refresh, network forward, argmax, crew sampling, where, env step} x ``n_ticks`` -- as ONE
        hipGraph (reference loop shape: visualize.py:547-582, one Python iteration per tick).  Everything a tick touches is
        device-resident (the env's step counter included: ``device_tick``), so ``graph.replay()`` advances the rollout by
        ``n_ticks`` ticks with a single launch from the host instead of ~20 eager launches per tick.

        Returns ``(graph, outputs)``; with ``record=True`` ``outputs`` holds static ``actions / rewards / done / truncated /
        obs_before`` tensors with a leading ``[n_ticks]`` dimension that every replay overwrites (parity tests).  Needs an env
        built with ``rng='philox', check_errors=False, export_state=False`` (both would synchronise inside the capture)."""
        env = self.env
        assert env.rng_kind == "philox", "graph replay needs the counter-based production stream"
        assert not env.check_errors and not env.export_state, "construct the env with check_errors=False, export_state=False"
        env.device_tick(True)
        B, A = env.batch, env.n_agents
        out: Dict[str, torch.Tensor] = {}
        if record:
            out = {"actions": torch.zeros(n_ticks, B, A, dtype=torch.int64, device=env.device),
                   "rewards": torch.zeros(n_ticks, B, A, dtype=torch.float32, device=env.device),
                   "done": torch.zeros(n_ticks, B, dtype=torch.bool, device=env.device),
                   "truncated": torch.zeros(n_ticks, B, dtype=torch.bool, device=env.device),
                   "obs_before": torch.zeros(n_ticks, *env.obs.shape, dtype=env.obs.dtype, device=env.device)}

        def tick(k):
            if record:
                out["obs_before"][k].copy_(env.obs)
            a, rew, done, trunc = self.tick()
            if record:
                out["actions"][k].copy_(a)
                out["rewards"][k].copy_(rew)
                out["done"][k].copy_(done)
                out["truncated"][k].copy_(trunc)

        cur = torch.cuda.current_stream(env.device)
        side = torch.cuda.Stream(env.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up on a side stream, as torch's capture rules ask (these ticks count: the rollout advances)
            for k in range(2):
                tick(k % n_ticks)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for k in range(n_ticks):
                tick(k)
        self.captured_warmup_ticks = 2
        return graph, out


class WindowedPolicyRollout:
    """Acting loop over a device-resident window of the last ``sequence_length`` flattened states per env.

    ``env`` must fuse the raw uint8 observation into step/reset (``obs=ObsConfig('raw', dtype=torch.uint8)``) and
    auto-reset.  ``featurizer`` is one of ``features.PerspectiveFeaturizer / GlobalFeaturizer / FlatFeaturizer`` built on
    the same env.  Per tick (train.py:345-383): ``featurizer.fit(window)``; agent *i* of env *b* acts by the imposter
    network if it is an imposter there, else by the crew network, on ITS view; dead agents get index 0 when
    ``mask_dead`` (train.py) and act like everybody else otherwise (visualize.py:547-560); with ``epsilon`` > 0 a
    uniformly random role-valid index replaces the greedy one with that probability (the env's sample_actions kernel
    supplies it).
    """

    def __init__(self, env, featurizer, imposter_model: nn.Module, crew_model: nn.Module, sequence_length: int = 2,
                 epsilon: float = 0.0, mask_dead: bool = True, seed: int = 0):
        assert env.obs_config.mode == "raw" and env.obs_config.dtype == torch.uint8 and env.auto_reset, (
            "construct the env with obs=ObsConfig('raw', dtype=torch.uint8), auto_reset=True")
        self.env, self.featurizer = env, featurizer
        self.imposter_model, self.crew_model = imposter_model, crew_model
        self.T, self.epsilon, self.mask_dead = int(sequence_length), float(epsilon), mask_dead
        self._gen = torch.Generator(device=env.device)
        self._gen.manual_seed(seed)
        self.window = None
        self._actions = torch.zeros(env.batch, env.n_agents, dtype=torch.int64, device=env.device)

    def reset(self):
        self.env.reset()
        self.window = self.env.obs.unsqueeze(1).repeat(1, self.T, 1)  # train.py:318-322: the first state T times
        return self.window

    @torch.no_grad()
    def act(self) -> torch.Tensor:
        env = self.env
        if self.window is None:
            self.reset()
        if not env.export_state:
            env.refresh_roles()
        self.featurizer.fit(self.window)
        imp_mask = env.imposter_mask  # [B, A]
        for i, (spatial, non_spatial) in enumerate(self.featurizer.generate_featurized_states()):
            q_imp = self.imposter_model(spatial, non_spatial).argmax(dim=1)
            q_crew = self.crew_model(spatial, non_spatial).argmax(dim=1)
            self._actions[:, i] = torch.where(imp_mask[:, i], q_imp, q_crew)
        if self.epsilon > 0.0:
            explore = torch.rand(self._actions.shape, device=env.device, generator=self._gen) <= self.epsilon  # train.py:359,371
            self._actions.copy_(torch.where(explore, env.sample_actions().to(torch.int64), self._actions))
        if self.mask_dead:
            self._actions.mul_(env.alive_agents.to(torch.int64) if env.export_state else self._alive_from_window())
        return self._actions

    def _alive_from_window(self) -> torch.Tensor:
        A = self.env.n_agents
        return self.window[:, -1, 2 * A:3 * A].to(torch.int64)  # flatten_state: alive flags follow the A (x, y) pairs

    @torch.no_grad()
    def step(self):
        """One tick: act, env.step, window update.  Returns (actions, rewards, done, truncated)."""
        env = self.env
        a = self.act()
        _, rew, done, trunc, _ = env.step(a)
        ended = (done | trunc).view(-1, 1, 1)
        nxt = torch.roll(self.window, shifts=-1, dims=1)  # train.py:388-389
        nxt[:, -1] = env.obs
        # an ended env was auto-reset inside the step: its window restarts from the fresh first state (train.py:452-457)
        self.window = torch.where(ended, env.obs.unsqueeze(1).expand(-1, self.T, -1), nxt)
        return a, rew, done, trunc

    @torch.no_grad()
    def run(self, n_steps: int, record: bool = False) -> Dict[str, torch.Tensor]:
        out: Dict[str, List[torch.Tensor]] = {"actions": [], "rewards": [], "done": [], "truncated": []}
        for _ in range(n_steps):
            a, rew, done, trunc = self.step()
            if record:
                for k, v in zip(out, (a, rew, done, trunc)):
                    out[k].append(v.clone())
        return {k: torch.stack(v) for k, v in out.items()} if record else {}
