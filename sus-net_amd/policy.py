"""Policy-in-the-loop rollout (BASELINE.json config 5).

The Q-networks themselves are out of scope as kernels: they are the reference's architectures run through
stock PyTorch-ROCm (`nn.Linear` -> hipBLASLt).  What this module adds is the loop around the HIP environment:
fused flat observation -> model -> argmax -> `step`, greedy as in the reference's `run_game`
(src/visualize.py:547-582), with every tensor staying on the device.

* ``MLP`` mirrors reference src/models/dqn.py:72-108 (`make_mlp` 322-329: Linear + PReLU, last activation
  dropped) INCLUDING the module names, so a reference checkpoint `{"state_dict", "config"}`
  (dqn.py:92-103) loads unchanged.  The reference ships no checkpoints (`.gitignore:2`), so benchmarks use a
  seeded random initialisation of the same architecture (layer dims of notebooks/experiment_1v1.ipynb cell 1).
* ``RandomEquiprobable`` mirrors dqn.py:111-138; in the loop a random crew is sampled by the environment's own
  `sample_actions` kernel (uniform over the role-valid indices, base.py:326-330).
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


def reference_imposter_mlp(env, components: Sequence[str], seed: int = 0) -> MLP:
    """`[F, 256, 128, 64, 16, n_imposter_actions]` (notebooks/experiment_1v1.ipynb cell 1), seeded init."""
    spec, o1, _ = env._make_obs(ObsConfig("flat", list(components)), 1)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        model = MLP([o1.shape[-1], 256, 128, 64, 16, env.n_imposter_actions])
    return model.to(env.device).eval()


class PolicyRollout:
    """Greedy policy-in-the-loop stepping of a batched env.

    Per tick: the flat observation (written by the previous step / reset kernel) feeds ``imposter_model`` and
    ``crew_model``; agent *i* of env *b* takes the imposter model's argmax if it is an imposter there, else the
    crew model's (all agents see the same flat features, as in the reference's FlatFeaturizer,
    model_ready.py:356-367).  ``crew_model=None`` = uniformly random crew via the env's sample_actions kernel.
    """

    def __init__(self, env, imposter_model: nn.Module, crew_model: Optional[nn.Module] = None,
                 components: Sequence[str] = ("onehot_pos",)):
        assert env.obs_config.mode == "flat" and list(env.obs_config.components) == list(components), (
            "construct the env with obs=ObsConfig('flat', components) so that step() fuses the observation")
        self.env, self.imposter_model, self.crew_model = env, imposter_model, crew_model
        B = env.batch
        self._spatial = torch.zeros(B, 1, 1, device=env.device)  # FlatFeaturizer's dummy spatial input
        self._actions = torch.zeros(B, env.n_agents, dtype=torch.int64, device=env.device)

    @torch.no_grad()
    def act(self) -> torch.Tensor:
        env = self.env
        if not env.export_state:
            env.refresh_roles()  # auto-reset may have re-drawn the imposter indices (base.py:273-278)
        feats = env.obs  # [B, F] float32, refreshed by reset()/step()
        a_imp = self.imposter_model(self._spatial, feats).argmax(dim=1)
        if self.crew_model is None:
            crew = env.sample_actions().to(torch.int64)  # [B, A] uniform role-valid (imposter slots overwritten)
        else:
            crew = self.crew_model(self._spatial, feats).argmax(dim=1).unsqueeze(1).expand(-1, env.n_agents)
        torch.where(env.imposter_mask, a_imp.unsqueeze(1), crew, out=self._actions)
        return self._actions

    @torch.no_grad()
    def run(self, n_steps: int, record: bool = False) -> Dict[str, torch.Tensor]:
        env = self.env
        out: Dict[str, List[torch.Tensor]] = {"actions": [], "rewards": [], "done": [], "truncated": []}
        for _ in range(n_steps):
            a = self.act()
            _, rew, done, trunc, _ = env.step(a)
            if record:
                out["actions"].append(a.clone())
                out["rewards"].append(rew.clone())
                out["done"].append(done.clone())
                out["truncated"].append(trunc.clone())
        return {k: torch.stack(v) for k, v in out.items()} if record else {}
