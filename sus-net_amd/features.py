"""Host-side mirror of the reference's sequence featurizers (src/features/model_ready.py) over the
observations the HIP kernels write.

The reference's ``fit(state_sequence[B, T, S])`` un-flattens every state in Python and loops over the batch
(model_ready.py:41-57).  Here ``fit(state_sequence)`` hands the ``[B, T, S]`` tensor of flattened states (a
window, a replay batch: any of uint8 / int32 / int64 / float32 / float64, on any device) to the HIP featurize
kernel (``env.featurize`` -> ``susnet_featurize``); ``fit()`` with no argument observes the env's CURRENT state
(T = 1) through the same writers.  The OUTPUT contract of ``generate_featurized_states()`` is the reference's:
one ``(spatial, non_spatial)`` pair per agent with the reference shapes and channel / column orders.

* ``FlatFeaturizer``        model_ready.py:309-370   -> ``(zeros[B, T, 1], feats[B, T, F])`` per agent
  (+ ``imposter_scent``: component.py:336-380, the one real-valued component: the ``susnet_scent`` kernel)
* ``GlobalFeaturizer``      model_ready.py:219-306   -> ``(spatial[B, T, A+2, N, N], [alive, job_status, onehot(agent)])``
* ``PerspectiveFeaturizer`` model_ready.py:82-216    -> per-agent channel rotation (self first) of the same planes
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from .env import ObsConfig


def imposter_scent(env, raw: torch.Tensor) -> torch.Tensor:
    """``ImposterScentFeaturizer`` (src/features/component.py:336-380) over flattened states ``raw[..., S]`` (any of the
    dtypes ``env.featurize`` accepts): ``[..., 4]`` float32 from the ``susnet_scent`` kernel -- the reference's float64
    quotients rounded to float32 and summed in float32, in agent order."""
    import ctypes as C

    from . import _lib as L

    rows = raw.to(env.device).reshape(-1, raw.shape[-1]).contiguous()
    assert rows.dtype in env._ROW_DTYPES, f"unsupported state dtype {rows.dtype}"
    out = torch.empty(rows.shape[0], 4, dtype=torch.float32, device=env.device)
    with torch.cuda.device(env.device):
        L.check(env.lib.susnet_scent(env._h, rows.data_ptr(), env._ROW_DTYPES[rows.dtype], rows.shape[0], out.data_ptr(), env._stream()))
    return out.reshape(*raw.shape[:-1], 4)


class FlatFeaturizer:
    """Concatenation of flat components (CompositeFeaturizer + FlatFeaturizer, model_ready.py:309-370).  Every component
    of src/features/component.py that works in the reference is accepted; all but ``"scent"`` are written by the HIP
    kernels' byte images, ``"scent"`` (real-valued, unused by the reference's notebooks) by its own small kernel."""

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
