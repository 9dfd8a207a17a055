"""Throughput of configurations OTHER than the four fully compiled-in BASELINE games: the byte-parallel family kernels (any job count up
to 8, 3..12 agents -- tagging 3..8 --, at most 3 imposters: susnet_family.h), the 1v1 wall-map duel kernel and the generic kernels behind
them. GPU box only.
Both trajectory layouts: separate tensors (actions, rewards, done, truncated, raw uint8 observation) and packed records."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("sus-net_amd")
B, T, reps = 65536, 128, 8
kw = dict(batch=B, auto_reset=True, export_state=False, check_errors=False)
cases = {
    "itg 1v1 walls (experiment_1v1.ipynb envs['Wall'])": lambda: pkg.BatchedImposterTrainingGround(1, 0, 0, -3, 0, 0, **kw),
    "itg 1v3 j2": lambda: pkg.BatchedImposterTrainingGround(3, 2, 0, -3, 1, 5, **kw),
    "itg 1v10 (visualizing_games.ipynb)": lambda: pkg.BatchedImposterTrainingGround(10, 0, 0, -3, 0, 0, **kw),
    "base 1v3 j5 (replay_buffer_test.ipynb)": lambda: pkg.BatchedFourRoomEnv(1, 3, 5, **kw),
    "base 1v2 j4 14x14 (cfg3: compiled in)": lambda: pkg.BatchedFourRoomEnv(1, 2, 4, grid_size=14, **kw),
    "base 1v2 j3 14x14": lambda: pkg.BatchedFourRoomEnv(1, 2, 3, grid_size=14, **kw),
    "base 2v6 j5 14x14": lambda: pkg.BatchedFourRoomEnv(2, 6, 5, grid_size=14, **kw),
    "base 3v9 j8 16x16 (12 agents, 3 imposters)": lambda: pkg.BatchedFourRoomEnv(3, 9, 8, grid_size=16, **kw),
    "base 3v5 j3 (three imposters)": lambda: pkg.BatchedFourRoomEnv(3, 5, 3, **kw),
    "base 4v9 j4 16x16 (13 agents: generic kernels)": lambda: pkg.BatchedFourRoomEnv(4, 9, 4, grid_size=16, **kw),
    "tagging 2v6 j4 14x14": lambda: pkg.BatchedFourRoomEnvWithTagging(2, 6, 4, grid_size=14, **kw),
    "tagging 1v4 j5 (tag5: compiled in)": lambda: pkg.BatchedFourRoomEnvWithTagging(1, 4, 5, **kw),
    "tagging 1v4 j4": lambda: pkg.BatchedFourRoomEnvWithTagging(1, 4, 4, **kw),
}
only = sys.argv[1:]
for name, make in cases.items():
    if only and not any(o in name for o in only):
        continue
    env = make()
    env.reset()
    line = f"{name:52s}"
    for packed in (False, True):
        if packed and env.record_layout() is None:
            line += "   packed: none"
            continue
        bufs = env.alloc_rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8), packed=packed)
        for _ in range(2):
            env.rollout_into(T, bufs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            env.rollout_into(T, bufs)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        line += f"   {'packed' if packed else 'separate'}: {ms*1e3/T:6.3f} us/tick {B*T/ms/1e6:7.2f} G"
        del bufs
    print(line, flush=True)
    del env
