"""Host-side mirror of the reference's sequence featurizers (src/features/model_ready.py) over the
observations the HIP kernels write.

The reference's ``fit(state_sequence[B, T, S])`` un-flattens every state in Python and loops over the batch
(model_ready.py:41-57).  Here ``fit(state_sequence)`` hands the ``[B, T, S]`` tensor of flattened states (a
window, a replay batch: any of uint8 / int32 / int64 / float32 / float64, on any device) to the HIP featurize
kernel (``env.featurize`` -> ``susnet_featurize``); ``fit()`` with no argument observes the env's CURRENT state
(T = 1) through the same writers.  The OUTPUT contract of ``generate_featurized_states()`` is the reference's:
one ``(spatial, non_spatial)`` pair per agent with the reference shapes and channel / column orders.

* ``FlatFeaturizer``        model_ready.py:309-370   -> ``(zeros[B, T, 1], feats[B, T, F])`` per agent
  (+ ``imposter_scent``: component.py:336-380, the one real-valued component, as device-tensor arithmetic)
* ``GlobalFeaturizer``      model_ready.py:219-306   -> ``(spatial[B, T, A+2, N, N], [alive, job_status, onehot(agent)])``
* ``PerspectiveFeaturizer`` model_ready.py:82-216    -> per-agent channel rotation (self first) of the same planes
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from .env import ObsConfig


def imposter_scent(env, raw: torch.Tensor) -> torch.Tensor:
    """``ImposterScentFeaturizer`` (src/features/component.py:336-380) over flattened states ``raw[..., S]``: for every
    alive agent other than agent 0, ``(N - dx) / N`` and ``(N - dy) / N`` (Python floats, i.e. float64) are accumulated
    into a float32 4-vector ``[x>0, x<=0, y>0, y<=0]`` in agent order.  Device-tensor arithmetic in exactly that
    order and precision (the only real-valued feature of the reference; the HIP writers carry small integers)."""
    A, N = env.n_agents, env.n_rows
    r = raw.to(torch.float64)
    x, y, alive = r[..., 0:2 * A:2], r[..., 1:2 * A:2], r[..., 2 * A:3 * A] != 0
    out = torch.zeros(*raw.shape[:-1], 4, dtype=torch.float32, device=raw.device)
    for i in range(1, A):
        xs = ((N - (x[..., i] - x[..., 0])) / N).to(torch.float32)
        ys = ((N - (y[..., i] - y[..., 0])) / N).to(torch.float32)
        zero = torch.zeros_like(xs)
        out[..., 0] += torch.where(alive[..., i] & (xs > 0), xs, zero)
        out[..., 1] += torch.where(alive[..., i] & ~(xs > 0), xs, zero)
        out[..., 2] += torch.where(alive[..., i] & (ys > 0), ys, zero)
        out[..., 3] += torch.where(alive[..., i] & ~(ys > 0), ys, zero)
    return out


class FlatFeaturizer:
    """Concatenation of flat components (CompositeFeaturizer + FlatFeaturizer, model_ready.py:309-370).  Every component
    of src/features/component.py that works in the reference is accepted; all but ``"scent"`` are written by the HIP
    kernels, ``"scent"`` (real-valued, unused by the reference's notebooks) is torch arithmetic on the same device."""

    def __init__(self, env, components: Sequence[str]):
        self.env = env
        self.components = list(components)
        hip = [c for c in self.components if c != "scent"]
        self.config = ObsConfig("flat", hip) if hip else None
        self.featurized_state = None

    @property
    def featurized_shape(self):
        n = 4 * self.components.count("scent")
        if self.config is not None:
            spec, o1, _ = self.env._make_obs(self.config, 1)
            n += o1.shape[-1]
        return 1, torch.tensor([n], dtype=torch.int)

    def fit(self, state_sequence=None) -> None:
        if state_sequence is not None:
            assert state_sequence.dim() == 3, "state_sequence is [B, T, S] (model_ready.py:44)"
        hip = None
        if self.config is not None:
            hip = (self.env.observe(self.config).unsqueeze(1) if state_sequence is None  # [B, T=1, F]
                   else self.env.featurize(state_sequence, self.config))               # [B, T, F]
        if "scent" not in self.components:
            self.featurized_state = hip
            return
        raw = (self.env.observe(ObsConfig("raw")).unsqueeze(1) if state_sequence is None
               else state_sequence.to(self.env.device))
        scent = imposter_scent(self.env, raw)
        parts, k = [], 0
        sizes = {c: self.env._make_obs(ObsConfig("flat", [c]), 1)[1].shape[-1] for c in set(self.components) - {"scent"}}
        for c in self.components:  # concatenation order = component order
            if c == "scent":
                parts.append(scent)
            else:
                parts.append(hip[..., k:k + sizes[c]])
                k += sizes[c]
        self.featurized_state = torch.cat(parts, dim=-1)

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


class PerspectiveFeaturizer:
    """model_ready.py:82-216.  The kernels write every agent's view directly (observation mode ``"persp"``: the agent
    channels in the order i, 0, .., i-1, i+1, .., A-1 -- what the reference's mutated order list reads after iteration i,
    model_ready.py:184-193 -- and the per-agent non-spatial blocks in that same order), so ``generate_featurized_states``
    only slices."""

    def __init__(self, env):
        self.env = env
        self.config = ObsConfig("persp")
        self.spatial = self.non_spatial = None  # [B, T, A, C, N, N], [B, T, A, F2]

    def fit(self, state_sequence=None) -> None:
        if state_sequence is None:
            sp, non = self.env.observe(self.config)
            self.spatial, self.non_spatial = sp.unsqueeze(1), non.unsqueeze(1)
        else:
            assert state_sequence.dim() == 3, "state_sequence is [B, T, S] (model_ready.py:44)"
            self.spatial, self.non_spatial = self.env.featurize(state_sequence, self.config)

    def generate_featurized_states(self):
        return [(self.spatial[:, :, i], self.non_spatial[:, :, i]) for i in range(self.env.n_agents)]
