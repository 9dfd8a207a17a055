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


def chase_parameters(model, n):
    """Parameter VALUES for the reference's `MLP([4 n, 8, 8, 8, 8, 6])` on the 1v1 `onehot_pos` features that make the imposter walk towards the
    crew member and KILL on its cell -- so that episodes END BY KILLS (with seeded random parameters every episode of a greedy run ends by
    truncation: the `done` rows of the ring would never be compared).  Layer 1 reads dx = x1 - x0, dy = y1 - y0 off the one-hots; layer 2 has
    PReLU slope -1 (|x|) and yields |dx|, |dy|, dx + 100, dy + 100 (positive: unchanged); layers 3, 4 have slope 1 (identity); the last layer
    forms Q = [STAY -0.5, UP dy, DOWN -dy, LEFT -dx, RIGHT dx, KILL 0.5 - |dx| - |dy|] (pred_prey.py:12-19, base.py:69-79).  Every value is a
    small integer or half-integer: exact in float32 in any summation order."""
    sd = {k: torch.zeros_like(v) for k, v in model.state_dict().items()}
    lin = [k for k in sd if k.endswith("weight") and sd[k].dim() == 2]
    slopes = [k for k in sd if k.endswith("weight") and sd[k].dim() == 1]
    assert len(lin) == 5 and len(slopes) == 4, list(sd)
    w1 = sd[lin[0]]
    for k in range(n):
        w1[0, 2 * n + k], w1[0, k] = float(k), -float(k)              # dx: agent 1's x one-hot minus agent 0's (component.py:226-240)
        w1[1, 3 * n + k], w1[1, n + k] = float(k), -float(k)          # dy
    w2, b2 = sd[lin[1]], sd[lin[1].replace("weight", "bias")]
    w2[0, 0] = w2[1, 1] = w2[2, 0] = w2[3, 1] = 1.0
    b2[2] = b2[3] = 100.0
    for k in (lin[2], lin[3]):
        for u in range(4):
            sd[k][u, u] = 1.0
    w5, b5 = sd[lin[4]], sd[lin[4].replace("weight", "bias")]
    b5[0] = -0.5
    w5[1, 3], b5[1] = 1.0, -100.0
    w5[2, 3], b5[2] = -1.0, 100.0
    w5[3, 2], b5[3] = -1.0, 100.0
    w5[4, 2], b5[4] = 1.0, -100.0
    w5[5, 0], w5[5, 1], b5[5] = -1.0, -1.0, 0.5
    for k, v in zip(slopes, (1.0, -1.0, 1.0, 1.0)):
        sd[k][...] = v
    model.load_state_dict(sd)


def collect(name, spec, hidden, trajectory_size, max_size, num_steps, seed, components=COMPONENTS, chase=False):
    env = make_env(spec)
    parts = {"onehot_pos": comp.OneHotAgentPositionFeaturizer, "alive_crew": comp.AliveCrewFeaturizer, "closest_crew": comp.ClosestAliveCrewFeaturizer}
    feat = FlatFeaturizer(env, comp.CompositeFeaturizer([parts[c](env) for c in components]))
    F = int(feat.featurized_shape[1][0])
    torch.manual_seed(seed)
    imposter_model = MLP([F, *([8, 8, 8, 8] if chase else hidden), env.n_imposter_actions]).eval()
    crew_model = MLP([F, *hidden, env.n_crew_actions]).eval()
    if chase:
        chase_parameters(imposter_model, env.n_cols)
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
                n_agents=int(env.n_agents), n_imposters=int(env.n_imposters), idx=int(buf.idx), size=int(n), components=list(components),
                imposter_dims=[F, *([8, 8, 8, 8] if chase else hidden), int(env.n_imposter_actions)], crew_dims=[F, *hidden, int(env.n_crew_actions)],
                episodes_ended=int(buf.dones[:n].sum()),
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
    # episodes that END BY KILLS: the 1v1 game of notebooks/experiment_1v1.ipynb, the imposter's network = chase_parameters
    # (walks to the crew member, kills on its cell), the crew's a seeded random MLP; `done` rows carry the terminal next_state
    itg = dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0, time_step_reward=0)  # (pred_prey.py:26-38 takes no max_time_steps: 1000)
    collect("itg_1v1_nowalls_kills_t1", dict(**{"class": "itg"}, kwargs=dict(itg, include_walls=False)), [48, 32, 32, 16], 1, 512, 420, 23,
            components=["onehot_pos"], chase=True)
    collect("itg_1v1_nowalls_kills_t1_wrap", dict(**{"class": "itg"}, kwargs=dict(itg, include_walls=False)), [32, 32, 16, 16], 1, 150, 333, 25,
            components=["onehot_pos"], chase=True)  # the ring wraps across episode ends
    # (on the wall map this chase ends at the first wall between the two and the episode runs its 1 000 steps: no fixture)


if __name__ == "__main__":
    main()
