"""N2 measurement: DeviceReplayBuffer.populate_fused = fused rollout launches + susnet_ring_append.
Prints transitions/s end to end and the ring kernel's own rate against its algorithmic bytes (the six reference tensors of a row)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("sus-net_amd")
from bench import CONFIGS, make_env  # noqa: E402


def main():
    out = []
    for cfg, Tw in (("cfg2", 2), ("cfg3", 2), ("cfg3", 5)):
        spec = CONFIGS[cfg]
        B = spec["batch"]
        env = make_env(pkg, spec, B, 7, 0, torch.device("cuda:0"))
        S, A, NI = env.flattened_state_size, env.n_agents, env.n_imposters
        ticks, per = 128, 32
        buf = pkg.DeviceReplayBuffer(max_size=ticks * B, state_size=S, trajectory_size=Tw, n_agents=A, n_imposters=NI, device=env.device)
        buf.populate_fused(env, ticks, ticks_per_launch=per)  # warm-up: one full trip round the ring (allocations, first launches, every page touched)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = buf.populate_fused(env, ticks, ticks_per_launch=per)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        row_bytes = 2 * Tw * S * 4 + A * 8 + A * 4 + 1 + 2 * NI  # states + next_states f32, actions i64, rewards f32, done, imposters i16
        out.append({"config": cfg, "batch": B, "trajectory_size": Tw, "state_size": S, "transitions_per_s": n / dt,
                    "ring_bytes_per_row": row_bytes, "ring_GBs_end_to_end": n * row_bytes / dt / 1e9,
                    "note": "end to end: env.reset + observe + window setup + 4 x (fused rollout of 32 ticks + susnet_ring_append)"})
        del buf, env
        torch.cuda.empty_cache()
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
