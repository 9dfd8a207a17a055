#!/usr/bin/env python3
"""Many-seed pin: per-step CRC32 streams of the UNMODIFIED reference env (container only).

    python tests/golden/generate_checksums.py        # rewrites tests/golden/crc_*.npz

SURVEY.md section 8c asks for >= 64 seeds per configuration; full traces of that many runs would be megabytes, so this
fixture stores, per configuration, seed and step, one CRC32 over everything the step produced (actions sampled,
positions, alive, completed jobs, reward bit patterns, done, truncated, the 13 info counters, voting state).  The
loop is the populate()-shaped one (replay_memory.py:96-143): reset(seed); {sample_actions; step; reset on end}.
`step_record_bytes` in tests/conftest.py defines the byte layout that is hashed; data only is committed.
"""
from __future__ import annotations

import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _refshim  # noqa: E402

_refshim.install()
from conftest import step_record_bytes  # noqa: E402
from generate_golden import GRID14, asym_grid, make_env  # noqa: E402
from src.metrics import SusMetrics  # noqa: E402

N_SEEDS, N_STEPS = 64, 96


def run(spec, seed):
    env = make_env(spec)
    tagging = spec["class"] == "tagging"
    np.random.seed(seed)
    env.reset()
    out = np.zeros(N_STEPS, dtype=np.uint32)
    for s in range(N_STEPS):
        a = env.sample_actions()
        state, rew, done, trunc, info = env.step(a)
        rec = step_record_bytes(
            actions=a, pos=env.agent_positions, alive=env.alive_agents, jobdone=env.completed_jobs,
            rewards=np.asarray(rew, dtype=np.float64), done=done, trunc=trunc,
            metrics=[info[m] for m in SusMetrics],
            used=env.used_tag_actions if tagging else None, counts=env.tag_counts if tagging else None,
            timer_left=state[6] if tagging else None)
        out[s] = zlib.crc32(rec)
        if done or trunc:
            env.reset()
    return out, env


def main():
    specs = {}
    for walls in (True, False):
        w = "w" if walls else "nw"
        for n_crew in (1, 2, 4, 7):
            specs[f"itg_1v{n_crew}_{w}"] = dict(**{"class": "itg"}, kwargs=dict(
                n_crew=n_crew, n_jobs=0 if n_crew == 1 else 2, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0 if n_crew == 1 else 6,
                time_step_reward=0, include_walls=walls))
        for (ni, nc), nj, rnd, shuf in (((1, 2), 4, True, True), ((1, 4), 1, True, False), ((2, 6), 4, True, True),
                                         ((3, 5), 0, False, True), ((1, 3), 4, False, False)):
            specs[f"base_{ni}v{nc}_j{nj}_{w}_o{int(rnd)}s{int(shuf)}"] = dict(**{"class": "base"}, kwargs=dict(
                n_imposters=ni, n_crew=nc, n_jobs=nj, is_action_order_random=rnd, shuffle_imposter_index=shuf, include_walls=walls,
                max_time_steps=60))
    specs["base14_2v6_j4"] = dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=6, n_jobs=4, max_time_steps=80),
                                  grid=GRID14.astype(int).tolist())
    specs["base13_asym_1v3_j3"] = dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=3, n_jobs=3, max_time_steps=50),
                                       grid=asym_grid(13, 5).astype(int).tolist())
    specs["tagging_1v4_j5_i6"] = dict(**{"class": "tagging"}, kwargs=dict(n_imposters=1, n_crew=4, n_jobs=5, tag_reset_interval=6))
    specs["tagging_2v5_j2_i9_tsr"] = dict(**{"class": "tagging"}, kwargs=dict(n_imposters=2, n_crew=5, n_jobs=2, tag_reset_interval=9,
                                                                            time_step_reward=-1, vote_reward=5, max_time_steps=70))
    for name, spec in specs.items():
        crcs = np.zeros((N_SEEDS, N_STEPS), dtype=np.uint32)
        for seed in range(N_SEEDS):
            crcs[seed], env = run(spec, 1000 + seed)
        meta = dict(spec)
        meta.update(seeds=[1000 + s for s in range(N_SEEDS)], n_steps=N_STEPS, n_agents=int(env.n_agents), n_jobs=int(env.n_jobs),
                    n_imposters=int(env.n_imposters), grid_used=np.asarray(env.grid, dtype=np.uint8).tolist())
        np.savez_compressed(os.path.join(HERE, f"crc_{name}.npz"), meta=np.array(json.dumps(meta)), crc=crcs,
                            crc_of_crcs=np.uint32(zlib.crc32(crcs.tobytes())))
        print(name, hex(zlib.crc32(crcs.tobytes())))


if __name__ == "__main__":
    main()
