#!/usr/bin/env python3
"""Golden vectors of the reference's ReplayBuffer.populate (container only).

    python tests/golden/generate_replay.py        # rewrites tests/golden/replay_*.npz

Runs the UNMODIFIED reference `ReplayBuffer(max_size, state_size, trajectory_size, n_agents, n_imposters).populate(env,
num_steps)` (src/replay_memory.py:11-143) on the unmodified reference env after `np.random.seed(seed)` and records the
buffer's tensors -- states / next_states [max, T, S], actions [max, A], rewards [max, A], dones [max, 1], imposters
[max, n_imp] -- plus idx and size.  Only data is committed; the native ring (susnet_ring_append) is compared with it on
the GPU box from the same numpy seed (tests/test_gpu_parity.py).
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
from generate_golden import GRID14, make_env  # noqa: E402
from src.replay_memory import ReplayBuffer  # noqa: E402

SPECS = {
    # name: (env spec, trajectory_size, max_size, num_steps, seed)
    "itg_1v1_nowalls_t2": (dict(**{"class": "itg"}, kwargs=dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                                                               time_step_reward=0, include_walls=False)), 2, 2048, 1500, 7),
    "base_1v2_j4_t3": (dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4, max_time_steps=40)), 3, 512, 420, 11),
    # ring smaller than the run: the buffer wraps (only the last max_size transitions survive, at rotated positions)
    "base_1v2_j4_t4_wrap": (dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4, max_time_steps=35)), 4, 200, 450, 12),
    "base14_2v6_j4_t1": (dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=6, n_jobs=4, max_time_steps=50),
                              grid=GRID14.astype(int).tolist()), 1, 400, 300, 13),
    "itg_1v3_j2_t3": (dict(**{"class": "itg"}, kwargs=dict(n_crew=3, n_jobs=2, kill_reward=-3, sabotage_reward=0, end_of_game_reward=6,
                                                          time_step_reward=-1, shuffle_imposter_index=True)), 3, 1024, 900, 14),
    "tagging_1v4_j5_t2": (dict(**{"class": "tagging"}, kwargs=dict(n_imposters=1, n_crew=4, n_jobs=5, tag_reset_interval=6, max_time_steps=45)),
                          2, 512, 330, 15),
}


def main():
    for name, (spec, T, max_size, num_steps, seed) in SPECS.items():
        env = make_env(spec)
        S = env.flattened_state_size
        buf = ReplayBuffer(max_size, S, T, env.n_agents, env.n_imposters)
        np.random.seed(seed)
        buf.populate(env, num_steps)
        n = buf.size
        meta = dict(spec)
        meta.update(seed=seed, trajectory_size=T, max_size=max_size, num_steps=num_steps, state_size=int(S), n_agents=int(env.n_agents),
                    n_imposters=int(env.n_imposters), idx=int(buf.idx), size=int(buf.size),
                    grid_used=np.asarray(env.grid).astype(int).tolist())
        states = buf.states[:n].numpy()
        nxt = buf.next_states[:n].numpy()
        assert np.array_equal(states, states.astype(np.int16)) and np.array_equal(nxt, nxt.astype(np.int16))
        out = os.path.join(HERE, f"replay_{name}.npz")
        np.savez_compressed(out, meta=json.dumps(meta), states=states.astype(np.int16), next_states=nxt.astype(np.int16),
                            actions=buf.actions[:n].numpy().astype(np.int16), rewards=buf.rewards[:n].numpy().astype(np.float32),
                            dones=buf.dones[:n].numpy().astype(np.uint8), imposters=buf.imposters[:n].numpy().astype(np.int16))
        print(name, "size", n, "idx", buf.idx, "episodes ended", int(buf.dones[:n].sum()), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
