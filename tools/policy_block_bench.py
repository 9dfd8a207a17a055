"""BASELINE config 5's tick, three ways on the same handle: one launch per tick (susnet_qnet_policy_step writing the replay feed), a 64-tick
block in ONE launch (susnet_qnet_policy_rollout), and the network kernel alone.  One JSON line; us per tick at 65 536 environments."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sus-net_amd")
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    T = 64
    comps = bench.POLICY_COMPONENTS
    env = bench.make_env(pkg, bench.CONFIGS["cfg5"], B, 1, 0, torch.device("cuda:0"), pkg.ObsConfig("flat", comps))
    env.reset()
    pol = pkg.PolicyRollout(env, pkg.policy.reference_imposter_mlp(env, comps, seed=0), None, components=comps, epsilon=0.1, mask_dead=True)
    net = pol.fused_imposter
    feed = env.alloc_feed(T)

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3

    for _ in range(3):  # warm-up: clocks, first launches
        env.policy_rollout_into(feed, T, net, epsilon=0.1, mask_dead=True)
    torch.cuda.synchronize()
    out = {"envs": B, "ticks_per_block": T}
    out["block_us_per_tick"] = timed(lambda: env.policy_rollout_into(feed, T, net, epsilon=0.1, mask_dead=True), 6) / (6 * T)
    k = [0]
    def tick():
        env.policy_tick_into(feed, k[0] % T, net, epsilon=0.1, mask_dead=True)
        k[0] += 1
    out["launch_per_tick_us"] = timed(tick, 4 * T) / (4 * T)
    out["network_alone_us"] = timed(lambda: env.qnet_forward(net), 100) / 100
    out["block_us_per_tick_again"] = timed(lambda: env.policy_rollout_into(feed, T, net, epsilon=0.1, mask_dead=True), 6) / (6 * T)
    out["block_of_5_us_per_tick"] = timed(lambda: env.policy_rollout_into(feed, 5, net, epsilon=0.1, mask_dead=True), 60) / (60 * 5)
    # both teams by their networks (round 5: the crew's image swapped in between the two passes of a tick)
    crew = pkg.policy.pack_mlp(env, pkg.policy.reference_crew_mlp(env, comps, seed=1), comps)
    for _ in range(2):
        env.policy_rollout_into(feed, T, net, epsilon=0.1, mask_dead=True, net_crew=crew)
    torch.cuda.synchronize()
    out["both_teams_block_us_per_tick"] = timed(lambda: env.policy_rollout_into(feed, T, net, epsilon=0.1, mask_dead=True, net_crew=crew), 6) / (6 * T)
    def tick2():
        env.policy_tick_into(feed, k[0] % T, net, net_crew=crew, epsilon=0.1, mask_dead=True)
        k[0] += 1
    out["both_teams_launch_per_tick_us"] = timed(tick2, 4 * T) / (4 * T)
    out["both_teams_block_of_5_us_per_tick"] = timed(lambda: env.policy_rollout_into(feed, 5, net, epsilon=0.1, mask_dead=True, net_crew=crew), 60) / (60 * 5)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
