"""Host-side mirror of the reference's sequence featurizers (src/features/model_ready.py) over the
observations the HIP kernels write.

The reference's ``fit(state_sequence[B, T, S])`` un-flattens every state in Python and loops over the batch
(model_ready.py:41-57).  Here ``fit(state_sequence)`` hands the ``[B, T, S]`` tensor of flattened states (a
window, a replay batch: any of uint8 / int32 / int64 / float32 / float64, on any device) to the HIP featurize
kernel (``env.featurize`` -> ``susnet_featurize``); ``fit()`` with no argument observes the env's CURRENT state
(T = 1) through the same writers.  The OUTPUT contract of ``generate_featurized_states()`` is the reference's:
one ``(spatial, non_spatial)`` pair per agent with the reference shapes and channel / column orders.

* ``FlatFeaturizer``        model_ready.py:309-370   -> ``(zeros[B, T, 1], feats[B, T, F])`` per agent
* ``GlobalFeaturizer``      model_ready.py:219-306   -> ``(spatial[B, T, A+2, N, N], [alive, job_status, onehot(agent)])``
* ``PerspectiveFeaturizer`` model_ready.py:82-216    -> per-agent channel rotation (self first) of the same planes
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from .env import ObsConfig


class FlatFeaturizer:
    def __init__(self, env, components: Sequence[str]):
        self.env = env
        self.config = ObsConfig("flat", list(components))
        self.featurized_state = None

    @property
    def featurized_shape(self):
        spec, o1, _ = self.env._make_obs(self.config, 1)
        return 1, torch.tensor([o1.shape[-1]], dtype=torch.int)

    def fit(self, state_sequence=None) -> None:
        if state_sequence is None:
            self.featurized_state = self.env.observe(self.config).unsqueeze(1)  # [B, T=1, F]
        else:
            assert state_sequence.dim() == 3, "state_sequence is [B, T, S] (model_ready.py:44)"
            self.featurized_state = self.env.featurize(state_sequence, self.config)  # [B, T, F]

    def generate_featurized_states(self) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        B, T = self.featurized_state.shape[:2]
        zeros = torch.zeros(B, T, 1, device=self.featurized_state.device)
        return [(zeros.clone(), self.featurized_state.clone()) for _ in range(self.env.n_agents)]  # model_ready.py:356-367


class GlobalFeaturizer:
    def __init__(self, env):
        self.env = env
        self.config = ObsConfig("planes")
        self.spatial = self.non_spatial = None

    def fit(self, state_sequence=None) -> None:
        if state_sequence is None:
            sp, non = self.env.observe(self.config)
            self.spatial, self.non_spatial = sp.unsqueeze(1), non.unsqueeze(1)  # [B, 1, C, N, N], [B, 1, A(+A)+J]
        else:
            assert state_sequence.dim() == 3, "state_sequence is [B, T, S] (model_ready.py:44)"
            self.spatial, self.non_spatial = self.env.featurize(state_sequence, self.config)  # [B, T, C, N, N], [B, T, .]

    def generate_featurized_states(self):
        A = self.env.n_agents
        B, T = self.spatial.shape[:2]
        out = []
        for agent_idx in range(A):  # model_ready.py:291-306
            onehot = torch.zeros(B, T, A, device=self.spatial.device)
            onehot[:, :, agent_idx] = 1
            out.append((self.spatial.clone(), torch.cat([self.non_spatial, onehot], dim=2)))
        return out


class PerspectiveFeaturizer(GlobalFeaturizer):
    @staticmethod
    def _orders(A: int, C: int, agent_idx: int):
        # model_ready.py:184-193 mutates ONE order list across the loop: after iteration i it reads
        # [i, 0, 1, ..., i-1, i+1, ...]
        channels = list(range(C))
        agents = list(range(A))
        channels[0] = agents[0] = agent_idx
        for k in range(1, agent_idx + 1):
            channels[k] = agents[k] = k - 1
        return channels, agents

    def generate_featurized_states(self):
        A = self.env.n_agents
        C = self.spatial.shape[2]
        out = []
        for agent_idx in range(A):  # model_ready.py:175-216
            ch, ag = self._orders(A, C, agent_idx)
            nb = (self.non_spatial.shape[2] - self.env.n_jobs) // A  # per-agent blocks: alive (, tag_counts)
            blocks = [self.non_spatial[:, :, k * A:(k + 1) * A][:, :, ag] for k in range(nb)]
            rest = self.non_spatial[:, :, nb * A:]
            out.append((self.spatial[:, :, ch].clone(), torch.cat(blocks + [rest], dim=2)))
        return out
