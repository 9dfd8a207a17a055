"""Long parity soak (GPU box): kernels vs the CPU oracle on the Philox stream for thousands of steps, default
max_time_steps (episodes reach the 1000-step truncation), step API and fused rollout interleaved."""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
pkg = importlib.import_module("sus-net_amd")
from oracle import oracle as om  # noqa: E402
import test_gpu_parity as T  # noqa: E402

om.build()
NAMES = sys.argv[1:] or ["itg_1v1_nowalls", "base_1v2_j4_14", "tagging_1v4_j5", "base_2v6_j4_14",
                         "tagging_2v6_j4_14", "itg_1v5_j3",  # (these two: byte-parallel FAMILY kernels -- run-time job count, roles drawn per episode)
                         "itg_1v1_walls", "itg_1v10", "base_3v9_j8_16"]  # (round 5: the wall-map duel kernel; 11 / 12 agents, three imposters)
for name in NAMES:
    B, seed = 1536, 5
    spec = dict(T.CONFIGS[name])
    env, ob = T.make_pair(pkg, om, name, B, seed, auto_reset=True, check_errors=False)
    env.reset()
    ob.reset(threads=0)
    steps = 0
    ended = 0
    for block in range(14):
        # 150 fused ticks ...
        traj = env.rollout(150)
        torch.cuda.synchronize()
        acts, rews, dones, truncs = (traj[k].cpu().numpy() for k in ("actions", "rewards", "done", "truncated"))
        for s in range(150):
            oa = ob.sample_actions()
            assert np.array_equal(acts[s], oa), (name, "actions", steps)
            orew, odone, otrunc, rc = ob.step(oa, threads=0)
            assert np.array_equal(rews[s].astype(np.float64).view(np.uint64), orew.view(np.uint64)), (name, "rewards", steps)
            assert np.array_equal(dones[s], odone.astype(bool)) and np.array_equal(truncs[s], otrunc.astype(bool)), (name, "flags", steps)
            e = (odone | otrunc).astype(bool)
            ended += int(e.sum())
            ob.reset(mask=e)
            steps += 1
        # ... then 20 drop-in steps
        for s in range(20):
            a = env.sample_actions().clone()
            oa = ob.sample_actions()
            assert np.array_equal(a.cpu().numpy(), oa), (name, "api actions", steps)
            _, rew, done, trunc, _ = env.step(a)
            orew, odone, otrunc, rc = ob.step(oa, threads=0)
            assert np.array_equal(rew.cpu().numpy().astype(np.float64).view(np.uint64), orew.view(np.uint64)), (name, "api rewards", steps)
            e = (odone | otrunc).astype(bool)
            ended += int(e.sum())
            ob.reset(mask=e)
            steps += 1
        env._export(full=True)
        T.compare_full_state(env, ob, f"{name} block {block}")
    ntr = int(env.lifetime_totals()[3])
    print(f"{name}: {steps} steps x {B} envs bit-exact; {ended} episodes, {ntr} truncated at max_time_steps")
    assert int(env.lifetime_totals()[0]) == ended
