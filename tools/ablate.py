"""Time the fused rollout under output ablations (which part of a tick costs what). GPU box only."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

pkg = importlib.import_module("sus-net_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
spec = bench.CONFIGS[cfg]
B, T, reps = spec["batch"], int(sys.argv[2]) if len(sys.argv) > 2 else 512, 16
for obs in (("raw", "none") if T > 128 else ("raw", "none", "flat", "planes")):  # float32 planes at 512 ticks would be a 44 GB buffer
    for store in (("actions", "rewards", "done", "truncated"), ()):
        env = bench.make_env(pkg, spec, B, 1234, 0, torch.device("cuda:0"))
        env.reset()
        oc = {"raw": pkg.ObsConfig("raw", dtype=torch.uint8), "none": None, "flat": pkg.ObsConfig("flat", ["onehot_pos"]),
              "planes": pkg.ObsConfig("planes")}[obs]
        bufs = env.alloc_rollout(T, store=store, obs=oc)
        for _ in range(3):
            env.rollout_into(T, bufs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            env.rollout_into(T, bufs)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{cfg} obs={obs:6s} store={'all' if store else 'none':4s}: {ms*1e3:8.1f} us/launch  {ms*1e3/T:6.3f} us/tick  "
              f"{B*T/ms/1e6:8.2f} G env-steps/s")
        del env, bufs
