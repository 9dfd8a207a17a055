"""Throughput of configurations that run the GENERIC (LDS-table) kernels. GPU box only."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("sus-net_amd")
B, T, reps = 65536, 128, 8
cases = {
    "itg 1v3 j2 (generic)": lambda: pkg.BatchedImposterTrainingGround(3, 2, 0, -3, 1, 5, batch=B, auto_reset=True, export_state=False, check_errors=False),
    "base 1v3 j5 (generic)": lambda: pkg.BatchedFourRoomEnv(1, 3, 5, batch=B, auto_reset=True, export_state=False, check_errors=False),
    "base 3v9 j8 16x16 (generic)": lambda: pkg.BatchedFourRoomEnv(3, 9, 8, batch=B, grid_size=16, auto_reset=True, export_state=False, check_errors=False),
    "tagging 2v6 j4 14x14 (generic)": lambda: pkg.BatchedFourRoomEnvWithTagging(2, 6, 4, batch=B, grid_size=14, auto_reset=True, export_state=False, check_errors=False),
    "tagging 1v4 j5 (compiled-in)": lambda: pkg.BatchedFourRoomEnvWithTagging(1, 4, 5, batch=B, auto_reset=True, export_state=False, check_errors=False),
}
for name, make in cases.items():
    env = make()
    env.reset()
    bufs = env.alloc_rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8))
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
    print(f"{name:34s} {ms*1e3/T:7.3f} us/tick  {B*T/ms/1e6:7.2f} G env-steps/s")
    del env, bufs
