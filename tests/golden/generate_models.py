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


if __name__ == "__main__":
    main()
