#!/usr/bin/env python3
"""Golden vectors for the Q-network mirrors (container only): the UNMODIFIED reference `SpatialDQN` / `MLP`
(src/models/dqn.py) are instantiated with seeded weights, run on seeded inputs, and their parameters (by
state_dict key), inputs and outputs are stored as data.

    python tests/golden/generate_models.py        # rewrites tests/golden/model_*.npz
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

from src.models.dqn import MLP, SpatialDQN  # noqa: E402


def dump(name, model, config, inputs):
    model.eval()
    with torch.no_grad():
        out = model(*inputs)
    arrays = {"param::" + k: v.detach().numpy() for k, v in model.state_dict().items()}
    arrays.update({f"input{i}": x.numpy() for i, x in enumerate(inputs)})
    arrays["output"] = out.numpy()
    path = os.path.join(HERE, f"model_{name}.npz")
    np.savez_compressed(path, meta=np.array(json.dumps({"config": config, "keys": list(model.state_dict().keys())})), **arrays)
    print(name, "params", len(model.state_dict()), "out", tuple(out.shape), "bytes", os.path.getsize(path))


def main():
    torch.manual_seed(0)
    cfg = dict(input_image_size=9, non_spatial_input_size=7, n_channels=[5, 8, 16], strides=[1, 1], paddings=[1, 1],
               kernel_size=[3, 3], dilations=[1, 1], rnn_layers=2, rnn_hidden_dim=32, rnn_dropout=0.0,
               mlp_hidden_layer_dims=[16, 8], n_actions=7)
    m = SpatialDQN(**cfg)
    sp = (torch.rand(3, 2, 5, 9, 9) < 0.1).float()
    ns = torch.randint(0, 2, (3, 2, 7)).float()
    dump("spatialdqn", m, cfg, (sp, ns))
    # strided / dilated variant: exercises calculate_cnn_output_dim and the repeated-last-layer rule
    cfg2 = dict(input_image_size=14, non_spatial_input_size=12, n_channels=[10, 12], strides=[2], paddings=[2],
                kernel_size=[3, 3], dilations=[2], rnn_layers=1, rnn_hidden_dim=24, rnn_dropout=0.0,
                mlp_hidden_layer_dims=[], n_actions=6)
    try:
        m2 = SpatialDQN(**cfg2)
        sp2 = (torch.rand(2, 3, 10, 14, 14) < 0.05).float()
        ns2 = torch.randint(0, 2, (2, 3, 12)).float()
        dump("spatialdqn_strided", m2, cfg2, (sp2, ns2))
    except Exception as exc:  # the reference's own size bookkeeping may reject the combination
        print("strided variant not representable in the reference:", type(exc).__name__, exc)
    cfg3 = dict(layer_dims=[20, 32, 16, 7])
    m3 = MLP(**cfg3)
    dump("mlp", m3, cfg3, (torch.zeros(4, 1, 1), torch.rand(4, 1, 20)))
    coordinate_mlp()


def coordinate_mlp():
    """The policy input of the reference's `no_wall_coord_features` / `wall_coord_features` experiments (notebooks/experiment_1v1.ipynb cell 1):
    `FlatFeaturizer` over `CoordinateAgentPositionsFeaturizer` (component.py:384-403: [x0, y0, x1, y1], not zeroed for a dead agent) feeding
    `MLP([4, 256, 128, 64, 16, n_actions])`.  States of a seeded random rollout of the 1v1 game on both maps (kills included: a dead crew
    member keeps its coordinates), their flattened form, the featurizer's rows and both teams' Q rows: what `susnet_qnet_forward` must
    reproduce from the state words (tests/test_gpu_parity.py)."""
    from generate_golden import make_env
    from src.features import component as comp
    from src.features.model_ready import FlatFeaturizer

    for tag, walls in (("nowalls", False), ("walls", True)):
        spec = {"class": "itg", "kwargs": dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0, time_step_reward=0,
                                               include_walls=walls)}
        env = make_env(spec)
        feat = FlatFeaturizer(env, comp.CoordinateAgentPositionsFeaturizer(env))
        F = int(feat.featurized_shape[1][0])
        torch.manual_seed(5 + int(walls))
        hidden = [48, 32, 32, 16]  # (the notebook's [256, 128, 64, 16] is compared with the torch module on the GPU box: a small fixture here)
        models = {"imposter": MLP([F, *hidden, env.n_imposter_actions]).eval(), "crew": MLP([F, *hidden, env.n_crew_actions]).eval()}
        with torch.no_grad():  # (default PReLU slopes are all 0.25 and default biases small: make every parameter matter)
            for m in models.values():
                for i, mod in enumerate(x for x in m.model if isinstance(x, torch.nn.PReLU)):
                    mod.weight.fill_((0.1, 0.3, 0.5, 0.7)[i])
        np.random.seed(77 + int(walls))
        state, _ = env.reset()
        rows, feats, q = [], [], {k: [] for k in models}
        for step in range(160):
            flat = env.flatten_state(state)
            feat.fit(torch.tensor(flat, dtype=torch.float64).reshape(1, 1, -1))
            spatial, non_spatial = next(iter(feat.generate_featurized_states()))
            rows.append(np.asarray(flat).astype(np.int16))
            feats.append(non_spatial.detach().reshape(-1).numpy().astype(np.float32))
            with torch.no_grad():
                for k, m in models.items():
                    q[k].append(m(spatial, non_spatial).reshape(-1).numpy())
            # a step WITHOUT a reset at the episode's end first (the terminal state -- a dead crew member -- is a row too), then reset
            state, _, done, trunc, _ = env.step(env.sample_actions())
            if done or trunc:
                flat = env.flatten_state(state)
                feat.fit(torch.tensor(flat, dtype=torch.float64).reshape(1, 1, -1))
                spatial, non_spatial = next(iter(feat.generate_featurized_states()))
                rows.append(np.asarray(flat).astype(np.int16))
                feats.append(non_spatial.detach().reshape(-1).numpy().astype(np.float32))
                with torch.no_grad():
                    for k, m in models.items():
                        q[k].append(m(spatial, non_spatial).reshape(-1).numpy())
                state, _ = env.reset()
        for step in range(24):  # ... and states with a DEAD crew member (direct assignment, as the trainer's callers may: base.py:397-402 aliases)
            state, _ = env.reset()
            env.alive_agents[1] = False
            flat = env.flatten_state((env.agent_positions, env.alive_agents))
            feat.fit(torch.tensor(flat, dtype=torch.float64).reshape(1, 1, -1))
            spatial, non_spatial = next(iter(feat.generate_featurized_states()))
            rows.append(np.asarray(flat).astype(np.int16))
            feats.append(non_spatial.detach().reshape(-1).numpy().astype(np.float32))
            with torch.no_grad():
                for k, m in models.items():
                    q[k].append(m(spatial, non_spatial).reshape(-1).numpy())
        rows = np.array(rows)
        arrays = {f"{k}::{name}": v.detach().numpy() for k, m in models.items() for name, v in m.state_dict().items()}
        meta = dict(spec, components=["coord_pos"], dims={k: [F, *hidden, int(n)] for k, n in
                                                       (("imposter", env.n_imposter_actions), ("crew", env.n_crew_actions))},
                    grid_used=np.asarray(env.grid).astype(int).tolist(), n_rows=int(len(rows)), dead_rows=int((rows[:, 4:6] == 0).any(axis=1).sum()))
        path = os.path.join(HERE, f"model_mlp_coord_1v1_{tag}.npz")
        np.savez_compressed(path, meta=np.array(json.dumps(meta)), rows=rows, features=np.array(feats), q_imposter=np.array(q["imposter"]),
                            q_crew=np.array(q["crew"]), **arrays)
        print("mlp_coord_1v1_" + tag, "rows", len(rows), "with a dead agent", meta["dead_rows"], "bytes", os.path.getsize(path))


if __name__ == "__main__":
    main()
