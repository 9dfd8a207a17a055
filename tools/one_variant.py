"""Run ONE rollout variant repeatedly (for rocprofv3 --pmc).  usage: one_variant.py cfg obs store reps"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

pkg = importlib.import_module("sus-net_amd")
cfg, obs, store, reps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
spec = bench.CONFIGS[cfg]
B, T = spec["batch"], 128
env = bench.make_env(pkg, spec, B, 1234, 0, torch.device("cuda:0"))
env.reset()
oc = {"raw": pkg.ObsConfig("raw", dtype=torch.uint8), "none": None, "flat": pkg.ObsConfig("flat", ["onehot_pos"]),
      "planes": pkg.ObsConfig("planes")}[obs]
bufs = env.alloc_rollout(T, store=(("actions", "rewards", "done", "truncated") if store == "all" else ()), obs=oc)
for _ in range(reps):
    env.rollout_into(T, bufs)
torch.cuda.synchronize()
