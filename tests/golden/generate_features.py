#!/usr/bin/env python3
"""Golden observation vectors from the UNMODIFIED reference featurizers (container only).

    python tests/golden/generate_features.py       # rewrites tests/golden/feat_*.npz

States are visited with the reference env under random actions; for each visited state the outputs of
`env.flatten_state` (base.py:234), every flat component featurizer (src/features/component.py) and the
Global / Perspective sequence featurizers (src/features/model_ready.py, T = 1) are stored as data.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402

from generate_golden import GRID14, make_env  # noqa: E402
from src.features import component as comp  # noqa: E402
from src.features.model_ready import GlobalFeaturizer, PerspectiveFeaturizer  # noqa: E402

FLAT_CLASSES = {
    "onehot_pos": comp.OneHotAgentPositionFeaturizer,
    "coord_pos": comp.CoordinateAgentPositionsFeaturizer,
    "alive_crew": comp.AliveCrewFeaturizer,
    "l1_crew": comp.L1CrewFeaturizer,
    "closest_crew": comp.ClosestAliveCrewFeaturizer,
    "walls3x3": comp.WallsFeaturizer,
    "dist_to_imp": comp.DistanceToImposterFeaturizer,
    "room_loc": comp.ImposterVSCrewRoomLocaionFeaturizer,
    "scent": comp.ImposterScentFeaturizer,
}


def collect(name, spec, seed, steps, flat_names, planes=True):
    env = make_env(spec)
    np.random.seed(seed)
    env.reset()
    rows = {k: [] for k in ["pos", "alive", "jobpos", "jobdone", "imp", "raw"]}
    feats = {k: [] for k in flat_names}
    sp, nsp, persp_sp, persp_nsp = [], [], [], []
    flat = {k: FLAT_CLASSES[k](env) for k in flat_names}
    gf = GlobalFeaturizer(env) if planes else None
    pf = PerspectiveFeaturizer(env) if planes else None
    for s in range(steps):
        state = (env.agent_positions, env.alive_agents,
                 *([env.job_positions, env.completed_jobs] if env.n_jobs > 0 else []))
        rows["pos"].append(env.agent_positions.copy())
        rows["alive"].append(env.alive_agents.copy())
        rows["jobpos"].append(env.job_positions.copy().reshape(-1, 2))
        rows["jobdone"].append(env.completed_jobs.copy())
        rows["imp"].append(env.imposter_mask.copy())
        rows["raw"].append(np.asarray(env.flatten_state(state), dtype=np.float64))
        for k, f in flat.items():
            feats[k].append(f.extract_features(state).numpy().astype(np.float32).reshape(-1))
        if planes:
            seq = torch.tensor(rows["raw"][-1]).reshape(1, 1, -1)
            gf.fit(seq)
            sp.append(gf.spatial[0, 0].numpy().copy())
            nsp.append(gf.non_spatial[0, 0].numpy().copy())
            pf.fit(seq)
            views = pf.generate_featurized_states()
            persp_sp.append(np.stack([v[0][0, 0].detach().numpy() for v in views]))
            persp_nsp.append(np.stack([v[1][0, 0].detach().numpy() for v in views]))
        a = env.sample_actions()
        _, _, done, trunc, _ = env.step(a)
        if done or trunc:
            env.reset()
    out = {k: np.array(v).astype(np.int16) if k != "raw" else np.array(v) for k, v in rows.items()}
    for k, v in feats.items():
        out["flat_" + k] = np.array(v, dtype=np.float32)
    if planes:
        out["planes_spatial"] = np.array(sp, dtype=np.float32)
        out["planes_non_spatial"] = np.array(nsp, dtype=np.float32)
        out["persp_spatial"] = np.array(persp_sp, dtype=np.float32)
        out["persp_non_spatial"] = np.array(persp_nsp, dtype=np.float32)
    meta = dict(spec)
    meta.update(seed=seed, flat=list(flat_names), n_agents=int(env.n_agents), n_jobs=int(env.n_jobs),
                n_imposters=int(env.n_imposters), grid_used=np.asarray(env.grid, dtype=np.uint8).tolist())
    path = os.path.join(HERE, f"feat_{name}.npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **out)
    print(name, "S=", steps, "size=", os.path.getsize(path))


ALL9 = list(FLAT_CLASSES)
NO_ROOM = [k for k in ALL9 if k != "room_loc"]


def main():
    itg = dict(n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0, time_step_reward=0)
    collect("itg_1v1_nowalls", dict(**{"class": "itg"}, kwargs=dict(itg, n_crew=1, include_walls=False)), 0, 300, ALL9, planes=False)  # J=0: reference planes featurizer raises
    collect("itg_1v3_walls", dict(**{"class": "itg"}, kwargs=dict(itg, n_crew=3)), 1, 300, ALL9, planes=False)
    collect("base_1v2_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4,
                                                               shuffle_imposter_index=False)), 2, 300, ALL9)
    collect("base14_1v2_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4,
                                                                 shuffle_imposter_index=False),
                                  grid=GRID14.astype(int).tolist()), 3, 300, NO_ROOM)
    # multi-imposter configs: only the components that do not index n_crew-sized vectors by agent-1
    collect("base_2v6_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=6, n_jobs=4)), 4, 200,
            ["onehot_pos", "coord_pos", "alive_crew", "walls3x3", "dist_to_imp", "room_loc", "scent"])
    collect("base14_2v6_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=6, n_jobs=4),
                                  grid=GRID14.astype(int).tolist()), 5, 200,
            ["onehot_pos", "coord_pos", "alive_crew", "walls3x3", "dist_to_imp", "scent"])


if __name__ == "__main__":
    main()
