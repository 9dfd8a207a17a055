"""Diagnostic: cycle shares of the fused rollout's per-tick segments (needs a -DSUSNET_STAMPS build loaded through
SUSNET_LIB_PATH).  Never quote this build's run time; read the SHARES."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

pkg = importlib.import_module("sus-net_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
spec = bench.CONFIGS[cfg]
env = bench.make_env(pkg, spec, spec["batch"], 1234, 0, torch.device("cuda:0"))
env.reset()
T, reps = 128, 8
packed = env.record_layout() is not None and "--separate" not in sys.argv
bufs = env.alloc_rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8), packed=packed)
env.rollout_into(T, bufs)
torch.cuda.synchronize()
env._state[16:16 + 128].zero_()
for _ in range(reps):
    env.rollout_into(T, bufs)
torch.cuda.synchronize()
seg = env._state[16:16 + 64].view(torch.int64).tolist()
swar = cfg in ("cfg3", "cfg4", "tag5")
names = (["next tick: words + sample + ranks", "step", "traj stores / roles", "episode end (reset)", "record / obs stores", "-", "-", "-"] if swar else
         ["preload", "sample(philox)", "action stores", "step_env", "done/trunc stores", "episode end", "obs flush+fill", "-"])
tot = sum(seg)
for n, v in zip(names, seg):
    print(f"{n:20s} {v / (T * reps):9.1f} ticks-of-s_memtime per tick  {100.0 * v / max(1, tot):5.1f} %")
print("total per tick", tot / (T * reps))
seg2 = env._state[80:80 + 64].view(torch.int64).tolist()
names2 = (["classes + destinations", "kills", "fix / sabotage", "tags + vote", "win + rewards", "-", "-", "-"] if swar else
          ["prologue(fresh,align,shuffle)", "moves (all agents)", "kills (all agents)", "fix/sabotage (all agents)", "tagging+win check", "rewards+trunc", "-", "-"])
for n, v in zip(names2, seg2):
    print(f"  step_env: {n:32s} {v / (T * reps):9.1f}")
