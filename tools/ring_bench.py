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

import ctypes as C  # noqa: E402

pkg = importlib.import_module("sus-net_amd")
L = importlib.import_module("sus-net_amd._lib")
from bench import CONFIGS, make_env  # noqa: E402

TILES = tuple(os.environ.get("RING_BENCH_VARIANTS", "default").split(","))  # "default", "tile=8", ...: the launch-shape test hook
KERNEL_LAUNCHES = 20


def main():
    out = []
    os.environ.pop("SUSNET_RING_TILE", None)
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
        # the append kernel by itself: one trajectory block, K launches between two HIP events (the ring position advances as in a run);
        # SUSNET_RING_TILE is read when the handle is created, so every tiling is a fresh env over the same trajectory shape
        kernel = {}
        for tile in TILES:
            os.environ.pop("SUSNET_RING_TILE", None)
            if tile != "default":
                os.environ["SUSNET_RING_TILE"] = tile.split("=")[1]
            env_t = make_env(pkg, spec, B, 7, 0, torch.device("cuda:0"))
            env_t.reset()
            raw8 = pkg.ObsConfig("raw", dtype=torch.uint8)
            bufs = env_t.alloc_rollout(per, obs=raw8, replay_feed=True)
            env_t.rollout_into(per, bufs)
            window = env_t.observe(raw8).unsqueeze(1).repeat(1, Tw, 1).contiguous()
            io = buf._ring_io(env_t, bufs, window)
            io.n_ticks = per
            idx = 0
            def launch():
                nonlocal idx
                io.idx = idx
                L.check(env_t.lib.susnet_ring_append(env_t._h, C.byref(io), env_t._stream()))
                idx = (idx + per * B) % buf.max_size
            for _ in range(3):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(KERNEL_LAUNCHES):
                launch()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / KERNEL_LAUNCHES
            kernel[tile] = {"us_per_launch": us, "written_TBs": per * B * row_bytes / us / 1e6}
            # ... and IN ITS PLACE in the pipeline: right after the rollout launch that wrote the trajectory block (which then still sits in
            # the 256 MB cache), each append bracketed by its own event pair -- what populate_fused / collect pay per block
            pairs = []
            for _ in range(KERNEL_LAUNCHES):
                env_t.rollout_into(per, bufs)
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record()
                launch()
                p1.record()
                pairs.append((p0, p1))
            torch.cuda.synchronize()
            us_p = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
            kernel[tile]["in_pipeline"] = {"us_per_launch_median": us_p[len(us_p) // 2], "us_per_launch_mean": sum(us_p) / len(us_p),
                                           "written_TBs_at_median": per * B * row_bytes / us_p[len(us_p) // 2] / 1e6, "launches": len(us_p)}
            del env_t, bufs, window, io
        os.environ.pop("SUSNET_RING_TILE", None)
        out[-1]["append_launch"] = {"ticks": per, "rows": per * B, "launches_timed": KERNEL_LAUNCHES, "by_variant": kernel,
                                    "note": "susnet_ring_append alone (k_ring_append[_tile] + k_ring_window) over one trajectory block; written = the six ring tensors"}
        del buf, env
        torch.cuda.empty_cache()
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
