#!/usr/bin/env python3
"""Golden vectors of the reference trainer's COLLECTION loop (container only).

    python tests/golden/generate_collect.py        # rewrites tests/golden/collect_*.npz

`src/train.py` cannot be imported here (pygame / ipywidgets are missing), so its acting-and-collecting loop, train.py:316-322 and
345-399 + 419-449, is restated below around the UNMODIFIED reference objects it drives: the env (`FourRoomEnv`), the featurizer
(`FlatFeaturizer` over a `CompositeFeaturizer` of `OneHotAgentPositionFeaturizer`, `AliveCrewFeaturizer`, `ClosestAliveCrewFeaturizer`),
the Q-networks (`MLP`) and the `ReplayBuffer`.  Greedy acting (epsilon = 0): every living imposter takes the argmax of the imposter
network, every living crew member the argmax of the crew network, dead agents keep index 0 (train.py:351-381); no training step, so
the networks never change.  Stored: the network parameters, the actions taken, and the ring's tensors after the run -- data only.  The
native path (`DeviceReplayBuffer.collect`: the policy tick writing the replay feed, then susnet_ring_append) is compared with it on the
GPU box from the same numpy seed (tests/test_gpu_parity.py).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402

from generate_golden import GRID14, make_env  # noqa: E402
from src.environment import StateFields  # noqa: E402
from src.features import component as comp  # noqa: E402
from src.features.model_ready import FlatFeaturizer  # noqa: E402
from src.models.dqn import MLP  # noqa: E402
from src.replay_memory import ReplayBuffer  # noqa: E402

COMPONENTS = ["onehot_pos", "alive_crew", "closest_crew"]  # the policy input of BASELINE config 5 (88 features on the 14x14 1v2 game)


def collect(name, spec, hidden, trajectory_size, max_size, num_steps, seed):
    env = make_env(spec)
    feat = FlatFeaturizer(env, comp.CompositeFeaturizer([comp.OneHotAgentPositionFeaturizer(env), comp.AliveCrewFeaturizer(env),
                                                         comp.ClosestAliveCrewFeaturizer(env)]))
    F = int(feat.featurized_shape[1][0])
    torch.manual_seed(seed)
    imposter_model = MLP([F, *hidden, env.n_imposter_actions]).eval()
    crew_model = MLP([F, *hidden, env.n_crew_actions]).eval()
    buf = ReplayBuffer(max_size, env.flattened_state_size, trajectory_size, env.n_agents, env.n_imposters)
    np.random.seed(seed)
    # ---- train.py:316-322
    state, _ = env.reset()
    state_sequence = np.zeros((buf.trajectory_size, buf.state_size))
    for i in range(buf.trajectory_size):
        state_sequence[i] = env.flatten_state(state)
    actions, margins = [], []
    for _ in range(num_steps):
        # ---- train.py:345-381 with eps = 0
        feat.fit(torch.tensor(state_sequence).unsqueeze(0))
        agent_actions = np.zeros(env.n_agents, dtype=np.int32)
        alive_agents = state[env.state_fields[StateFields.ALIVE_AGENTS]]
        with torch.no_grad():
            for agent_idx, (spatial, non_spatial) in enumerate(feat.generate_featurized_states()):
                if not alive_agents[agent_idx]:
                    continue
                q = (imposter_model if env.imposter_mask[agent_idx] else crew_model)(spatial, non_spatial).reshape(-1)
                agent_actions[agent_idx] = int(torch.argmax(q))
                top = torch.topk(q, 2).values
                margins.append(float(top[0] - top[1]))
        # ---- train.py:383-399
        next_state, reward, done, trunc, info = env.step(agent_actions=agent_actions)
        next_state_sequence = np.roll(state_sequence.copy(), -1, axis=0)
        next_state_sequence[-1] = env.flatten_state(next_state)
        buf.add(state=state_sequence, action=agent_actions, reward=reward, done=done, next_state=next_state_sequence, imposters=env.imposter_idxs)
        actions.append(agent_actions.copy())
        # ---- train.py:419-449
        if done or trunc:
            state, _ = env.reset()
            state_sequence = np.zeros((buf.trajectory_size, buf.state_size))
            for i in range(buf.trajectory_size):
                state_sequence[i] = env.flatten_state(state)
        else:
            state = next_state
            state_sequence = next_state_sequence
    n = buf.size
    meta = dict(spec)
    meta.update(seed=seed, trajectory_size=trajectory_size, max_size=max_size, num_steps=num_steps, state_size=int(env.flattened_state_size),
                n_agents=int(env.n_agents), n_imposters=int(env.n_imposters), idx=int(buf.idx), size=int(n), components=COMPONENTS,
                imposter_dims=[F, *hidden, int(env.n_imposter_actions)], crew_dims=[F, *hidden, int(env.n_crew_actions)],
                smallest_argmax_margin=min(margins), grid_used=np.asarray(env.grid).astype(int).tolist())
    arrays = {"imposter::" + k: v.numpy() for k, v in imposter_model.state_dict().items()}
    arrays.update({"crew::" + k: v.numpy() for k, v in crew_model.state_dict().items()})
    out = os.path.join(HERE, f"collect_{name}.npz")
    np.savez_compressed(out, meta=json.dumps(meta), taken=np.array(actions, dtype=np.int16), states=buf.states[:n].numpy().astype(np.int16),
                        next_states=buf.next_states[:n].numpy().astype(np.int16), actions=buf.actions[:n].numpy().astype(np.int16),
                        rewards=buf.rewards[:n].numpy().astype(np.float32), dones=buf.dones[:n].numpy().astype(np.uint8),
                        imposters=buf.imposters[:n].numpy().astype(np.int16), **arrays)
    print(name, "size", n, "idx", buf.idx, "episodes ended", int(buf.dones[:n].sum()), "smallest argmax margin", min(margins),
          os.path.getsize(out), "bytes")


def main():
    base14 = dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4, shuffle_imposter_index=False, max_time_steps=60),
                  grid=GRID14.astype(int).tolist())
    # (trajectory_size 1: the reference MLP flattens the whole window into its input -- dqn.py:84-88 -- and the native Q-network kernel
    # reads the current state; longer windows go through WindowedPolicyRollout's torch modules)
    collect("base14_1v2_j4_t1", base14, [48, 32, 32, 16], 1, 512, 420, 21)
    collect("base14_1v2_j4_t1_wrap", base14, [64, 32, 16, 16], 1, 200, 330, 22)  # the ring wraps: only the last 200 transitions survive


if __name__ == "__main__":
    main()
