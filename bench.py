#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Sus-Net environment hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4] [--mode fused|step]

One pass of the hot path = every env samples uniform role-valid actions, steps, and auto-resets on done|truncated:
the random-action rollout BASELINE.json names (`ReplayBuffer.populate`'s loop, reference
src/replay_memory.py:96-143, minus the buffer).

  --mode fused (default): a bench STEP is ONE `susnet_rollout` launch = `--ticks` (default 512) lockstep ticks over
                          the whole batch, the trajectory (actions, rewards, done, truncated, raw observation)
                          written to HBM every tick.  `--steps K --warmup W` = W untimed launches, then K timed ones;
                          `value` = batch x ticks x K / wall time of the K launches, and `roofline` is derived from
                          HIP events around the SAME K launches (one number, one set of launches).
  --mode step           : the drop-in API; a bench step is one tick = two launches (`sample_actions` + `step`).

The JSON line: `value` = env-steps/s of the named configuration (default cfg2 = BASELINE.json configs[1]).  `roofline`:
`achieved` / `frac` are computed from the bytes a launch actually has to move -- the trajectory records it stores (the state
of a fused rollout never leaves the chip; the committed PMC pass measures the same bytes: `traffic`); the figure SURVEY.md
section 8(d) defines for the one-step API (12A + 4J + 6 bytes per env-step, state read + written every step) is reported next
to it as `survey_8d_*`.  `other_configs` holds the same measurement (5 warm-up + 20 timed launches) for the other BASELINE
configurations: cfg3, cfg4 (the 8-GPU shard: 32 768 envs), tag5 and cfg5 (policy in the loop: eager and as a replayed
hipGraph).

N > 1: one rank per GPU (torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Started as a
plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) the script launches the N ranks itself, before
it touches the GPU.  The batch is sharded by contiguous global env ids, per-GPU batch fixed (weak scaling); the only
collective is one all-gather of the episode metrics after the timed region (sus-net_amd/dist.py).  Rank 0 prints ONE
JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# kernel arguments in device memory (see sus-net_amd/__init__.py); must be in the environment before HIP initialises
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(workload="cfg2: ImposterTrainingGround 1v1, 9x9 no walls, 0 jobs (notebook rewards), batch 65536/GPU",
                 cls="itg", kw=dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                                    time_step_reward=0, include_walls=False), n=9, A=2, J=0, batch=65536),
    # the reference's own 1v1 experiments run on its four-room WALL map (include_walls=True is the constructor default, base.py:119;
    # notebooks/experiment_1v1.ipynb envs['Wall']): cfg2 with the walls in
    "cfg2w": dict(workload="cfg2w: ImposterTrainingGround 1v1, 9x9 four-room wall map, 0 jobs (notebook rewards), batch 65536/GPU",
                  cls="itg", kw=dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                                     time_step_reward=0, include_walls=True), n=9, A=2, J=0, batch=65536),
    "cfg3": dict(workload="cfg3: FourRoomEnv 1v2, 14x14 walled, 4 jobs, batch 65536/GPU",
                 cls="base", kw=dict(n_imposters=1, n_crew=2, n_jobs=4), n=14, A=3, J=4, batch=65536),
    "cfg4": dict(workload="cfg4: FourRoomEnv 2v6, 14x14 walled, 4 jobs, batch 32768/GPU",
                 cls="base", kw=dict(n_imposters=2, n_crew=6, n_jobs=4), n=14, A=8, J=4, batch=32768),
    "tag5": dict(workload="tag5: FourRoomEnvWithTagging 1v4, 9x9 walled, 5 jobs, votes every 50 steps (notebooks/experiment.ipynb), batch 65536/GPU",
                 cls="tagging", kw=dict(n_imposters=1, n_crew=4, n_jobs=5), n=9, A=5, J=5, batch=65536),
    # BASELINE.json configs[4]: cfg3's env driven by the reference-architecture MLP (no checkpoints ship with the
    # reference: seeded random init), greedy imposter + uniformly random crew, everything on the device
    "cfg5": dict(workload="cfg5: cfg3 env (1v2, 14x14 walled, 4 jobs) driven by MLP[88,256,128,64,16,7] imposter policy "
                          "(float32 on the f32-input MFMA) + random crew, the whole tick one HIP kernel (k_qnet_step; ticks per launch: see "
                          "config.policy_forward), batch 65536/GPU",
                 cls="base", kw=dict(n_imposters=1, n_crew=2, n_jobs=4), n=14, A=3, J=4, batch=65536, policy=True),
}
POLICY_COMPONENTS = ["onehot_pos", "alive_crew", "closest_crew"]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA, dense (= the f32 vector rate)


def algorithmic_bytes_per_step(A, J, N, obs):
    """B_step = 12A + 4J + 6 (+ 4*A*2N flat one-hot, + 4*(A+2)*N^2 planes)."""
    b = 12 * A + 4 * J + 6
    if obs == "flat":
        b += 4 * A * 2 * N
    elif obs == "planes":
        b += 4 * (A + 2) * N * N
    return b


def stored_bytes_per_step(A, J, N, obs, raw_size=None, record_bytes=None):
    """What the fused rollout writes per env-step: actions u8 + rewards f32 + done + truncated + the observation.
    raw_size: flattened_state_size of the configuration (3A + 3J, + 2A + 1 with tagging); record_bytes: the packed record."""
    if record_bytes:
        return int(record_bytes)
    b = A + 4 * A + 2
    if obs == "raw":
        b += raw_size if raw_size is not None else 3 * A + (3 * J if J else 0)
    elif obs == "flat":
        b += 4 * A * 2 * N
    elif obs == "planes":
        b += 4 * (A + 2) * N * N + 4 * (A + J)
    return b


def profiled_traffic(config, obs, packed, batch, ticks, bytes_per_step=None):
    """HBM bytes per bench step of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_pmc_summary.json, written by tools/summarize_profiles.py: one entry per config, separate --pmc runs for
    FETCH_SIZE and WRITE_SIZE, KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  bench.py cannot
    collect PMCs itself.  An entry only counts for the run it was profiled on: same config, trajectory layout, batch, ticks."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        d = d.get(config, d if config == "cfg2" else {})  # (round-1 summaries held cfg2 only, at top level)
        cmd = d.get("command")  # (round >= 3 summaries: what the profiled command was)
        if cmd is not None and obs == "raw" and (bool(cmd.get("packed")) != bool(packed) or cmd.get("batch") != batch or cmd.get("ticks") != ticks):
            return None, None
        if cmd is not None and cmd.get("bytes_per_env_step") not in (None, bytes_per_step):  # (another record format of the same configuration)
            return None, None
        if cmd is None and obs == "raw" and not packed:  # (older summaries profiled the packed default)
            return None, None
        wkey = {"raw": "pmc_WRITE_SIZE", "planes": "pmc_W_planes", "flat": "pmc_W_flat"}.get(obs)
        # (per bench step: a step whose record array would pass 2 GiB runs as several consecutive launches)
        pick = lambda grp, ctr: next(v[ctr]["mean_per_launch"] * v.get("launches_per_bench_step", 1) for k, v in d[grp].items() if "k_rollout" in k)
        w = pick(wkey, "WRITE_SIZE")
        f = pick("pmc_FETCH_SIZE", "FETCH_SIZE") if obs == "raw" else 0.0
        return (w + 2.0 * f) * 1024.0, os.path.basename(files[-1])
    except (KeyError, StopIteration, ValueError, TypeError):
        return None, None


KERNEL_NAMES = {"cfg2": "k_rollout_duel<PhiloxRng, OUT, false>", "cfg2w": "k_rollout_duel<PhiloxRng, OUT, true>", "cfg3": "k_rollout_swar<Spec<3,4,..>, OUT, PhiloxRng>",
                "cfg4": "k_rollout_swar2<Spec<8,4,..>, OUT, PhiloxRng>", "tag5": "k_rollout_swar<Spec<5,5,2,..>, OUT, PhiloxRng>"}


def roofline_block(config, spec, B, ticks, obs, packed, raw_size, record_bytes, avg_launch_s, pair_us=None, with_traffic=True):
    """The dominant kernel's roofline figures for one fused launch of `ticks` ticks over B envs."""
    A, J, N = spec["A"], spec["J"], spec["n"]
    steps = B * ticks
    b_stored = stored_bytes_per_step(A, J, N, obs, raw_size, record_bytes if packed else None)
    b_8d = algorithmic_bytes_per_step(A, J, N, obs)
    achieved = steps * b_stored / avg_launch_s / 1e9
    traffic, src = profiled_traffic(config, obs, packed, B, ticks, b_stored) if with_traffic else (None, None)
    kernel = KERNEL_NAMES.get(config, "k_rollout")
    if config == "cfg4" and B != 32768:
        kernel = "k_rollout_swar<Spec<8,4,..>, OUT, PhiloxRng>"
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_unit": "bytes per bench step (PMC: WRITE_SIZE + 2 x FETCH_SIZE)", "traffic_source": src,
        "traffic_frac": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "bytes_per_env_step": b_stored, "bytes_per_launch": steps * b_stored,
        "survey_8d_bytes_per_env_step": b_8d, "survey_8d_achieved": steps * b_8d / avg_launch_s / 1e9,
        "survey_8d_frac": steps * b_8d / avg_launch_s / 1e9 / HBM_PEAK_GBS,
        "kernel": kernel, "env_steps_per_launch": steps, "avg_launch_us": avg_launch_s * 1e6, "avg_launch_us_event_pairs": pair_us,
        "note": "achieved / frac = bytes the launch stores (trajectory records: the state of a fused rollout stays on chip) / "
                "avg_launch_us / 8 TB/s; survey_8d_* = the same with SURVEY.md 8(d)'s 12A + 4J + 6 bytes per env-step (the one-step "
                "API's state read + write, which a fused rollout does not perform: r01 / r02 lines reported this figure as frac); "
                "avg_launch_us = HIP-event time of the timed region (events on the launch stream around the K timed launches) / K; "
                "avg_launch_us_event_pairs = mean of 8 further launches bracketed one by one (cross-check, outside the timed "
                "region); traffic = PMC bytes of the committed rocprofv3 pass of this command and layout (profiles/), null when "
                "none matches",
    }


def make_env(pkg, spec, batch, seed, env_id_base, device, obs_cfg=None):
    cls = {"itg": pkg.BatchedImposterTrainingGround, "base": pkg.BatchedFourRoomEnv,
           "tagging": pkg.BatchedFourRoomEnvWithTagging}[spec["cls"]]
    kw = dict(spec["kw"])
    walls = kw.pop("include_walls", True)
    return cls(**kw, grid=pkg.four_room_grid(spec["n"], walls), batch=batch, device=device, rng="philox", seed=seed,
               env_id_base=env_id_base, auto_reset=True, export_state=False, check_errors=False, obs=obs_cfg)


def cpu_baseline(spec, target_seconds=12.0):
    """The CPU oracle (oracle/susnet_oracle.c, proven step-for-step equal to the reference on the golden
    traces) timed on this box's host cores with the same loop shape. Reported, never the target."""
    import numpy as np
    from oracle import oracle as om

    kw = dict(spec["kw"])
    walls = kw.pop("include_walls", True)
    pkg = importlib.import_module("sus-net_amd")
    grid = pkg.four_room_grid(spec["n"], walls).astype(np.uint8)
    if spec["cls"] == "itg":
        kw.setdefault("shuffle_imposter_index", False)
    cfg = om.make_config(spec["cls"], grid=grid, **kw)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    B = 64 * cores
    ob = om.OracleBatch(cfg, B)
    ob.seed_mt(range(B))  # the reference's own stream: numpy-legacy MT19937, one seed per env
    out = {}
    for label, threads in (("1", 1), ("all", cores)):
        steps = 2000
        t0 = time.perf_counter()
        n, _, _ = ob.random_rollout(steps, threads)
        dt = time.perf_counter() - t0
        rate = n / dt
        steps = max(2000, int(rate * (target_seconds / 2) / B))
        t0 = time.perf_counter()
        n, eps, _ = ob.random_rollout(steps, threads)
        dt = time.perf_counter() - t0
        out[label] = dict(rate=n / dt, n=n, seconds=dt, episodes=eps)
    return dict(
        value=out["all"]["rate"], unit="env-steps/s", cores=cores, kind="port",
        kind_note=("the C restatement of the reference algorithm (oracle/, pinned to the reference by the golden fixtures), OpenMP over "
                   "independent envs; the reference itself is Python and never travels to the GPU box: its figures below are constants "
                   "measured once in the build container, not on this host"),
        sample=(f"C oracle, numpy-legacy MT19937 stream, {B} envs x {out['all']['n'] // B} steps random-action rollout "
                f"with reset on done|truncated ({out['all']['seconds']:.1f} s on {cores} threads)"),
        single_thread_value=out["1"]["rate"],
        reference_python_measured_in_build_container=dict(
            note="reference Python env, Xeon 2.1 GHz, 1 process / 8 processes (BASELINE.md section 2); not measured on this box",
            cfg2=[20.7e3, 158e3], cfg3_9x9=[12.8e3, 96e3], cfg4_9x9=[6.8e3, 53e3]),
    )


def config_block(spec, mode, obs, B, world, ticks_per_step, layout, agent_steps_per_s):
    """The line's `config`: the workload BASELINE.json names, the per-GPU and whole-job batch, the data-parallel degree."""
    return {"workload": spec["workload"], "mode": mode, "obs": obs, "batch_per_gpu": B,
            "global_batch": B * world, "ticks_per_launch": ticks_per_step,
            "step_definition": (f"one susnet_rollout launch = {ticks_per_step} lockstep ticks x {B} envs per GPU" if mode == "fused"
                                else "one lockstep tick (sample_actions + step)"),
            "env_steps_per_bench_step": B * ticks_per_step * world,
            "trajectory_layout": layout,
            "rng": "philox4x32-10 in-kernel", "auto_reset": True, "parallelism": f"dp{world}",
            "agent_steps_per_s": agent_steps_per_s}


def dry_run(args, pkg, rank, world):
    """--dry-run: the distributed control flow of a real run with no device behind it (see the flag's help)."""
    import torch
    import torch.distributed as dist

    spec = CONFIGS[args.config]
    B = args.batch or spec["batch"]
    mode = "policy" if spec.get("policy") else args.mode
    ticks_per_step = args.ticks if mode == "fused" else 1
    start, count = pkg.dist.shard_range(B * world, rank, world)  # the env ids this rank's handle would own (env_id_base = rank * B)
    assert (start, count) == (rank * B, B)
    if world > 1:
        dist.barrier()                                   # sync_all() before the timed region
        tmax = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)      # the MAX over ranks of the timed region
        assert tmax.item() == world - 1
        mine = torch.tensor([float(start)], dtype=torch.float64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)                      # per-rank launch times in the real run; here: every rank's first env id
        starts = [int(v.item()) for v in allv]
    else:
        starts = [start]
    table = pkg.dist.all_gather_totals(torch.full((12,), rank + 1, dtype=torch.int64))  # the ONE collective of the path (node metrics)
    assert table.shape == (world, 12) and int(table[:, 0].sum()) == world * (world + 1) // 2
    line = {"metric": "env-steps/s", "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "dry_run": True, "shard_first_env_ids": starts,
            "config": config_block(spec, mode, args.obs, B, world, ticks_per_step, None, None)}
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without the torch.distributed.run environment: start the N ranks ourselves (one per
    GPU), BEFORE this process makes any GPU call, relay their output and return their exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(launch_command(n_gpus, port, sys.argv[1:]), env=env)


def launch_command(n_gpus: int, port: int, argv):
    """The command `self_launch` runs: the driver's own contract (torch.distributed.run, one rank per GPU, 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def check_world(world: int, gpus: int):
    """bench.py never reports a line for a GPU count other than the one it was asked for."""
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but the launched world has {world} rank(s) (WORLD_SIZE); refusing to report "
                         f"a line for a different GPU count")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="timed bench steps (fused: launches of --ticks ticks; step mode: ticks)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="fused", choices=["fused", "step"])
    ap.add_argument("--obs", default="raw", choices=["none", "raw", "flat", "planes"])
    ap.add_argument("--ticks", type=int, default=512, help="ticks per fused launch (= per bench step in fused mode)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--packed", type=int, default=-1, help="fused + raw obs: 1 = one packed record per env-step (compiled-in configurations: "
                    "the same fields, one or a few wide stores per lane instead of one narrow store per tensor), 0 = separate "
                    "trajectory tensors, 2 = the COMPACT record where the configuration has one (cfg2: 16 bytes, actions and flags in one "
                    "byte), -1 (default) = compact, else packed, where the configuration has it")
    ap.add_argument("--settle-ms", type=float, default=150.0, help="before the W warm-up steps: untimed launches of the same bench step until the device "
                    "has been busy this long.  An MI355X that has been idle takes tens of milliseconds of sustained load to reach its steady clocks: 5 "
                    "warm-up launches of 0.1-0.3 ms each leave the timed region on the ramp (cfg3: 107 G with 5 warm-up launches, 118.5 G with 50 "
                    "or 400 -- same box, same binary).  Reported in the line as `settle_ms`; 0 = none")
    ap.add_argument("--policy-block", type=int, default=5, help="cfg5: ticks per launch of the policy loop (susnet_qnet_policy_rollout: the network image and the "
                    "launch are paid once per block, the weights are fixed within it).  Default 5 = the reference trainer's train_step_interval (train.py:295, "
                    "402-416: the networks change every 5 env steps), i.e. its acting loop between two optimizer steps; 64 = run_game's loop with fixed networks "
                    "(visualize.py:547-582; reported next to the headline as `policy_forms`); 0 = one launch per tick (susnet_qnet_policy_step)")
    ap.add_argument("--repeats", type=int, default=5, help="the K-launch timed region is run this many more times back to back (after the "
                    "headline measurement, which stays as it is): median / min / max of the repeats are reported in `repeats`")
    ap.add_argument("--dry-run", action="store_true", help="rehearse the N-rank launch on a box without N GPUs: the ranks start, rendezvous and run every "
                    "collective of the real run (world check, barrier, max over ranks, the metrics all-gather) on host tensors, make NO device call, and rank 0 "
                    "prints the line with its config block as the real run builds it and `value` null, `dry_run` true (tests/test_dist_gloo.py: BASELINE "
                    "config 4 = --gpus 8 --config cfg4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the step-API leg, the observation sweep and other_configs")
    args = ap.parse_args()
    assert args.gpus >= 1 and args.steps >= 1 and args.warmup >= 0 and args.ticks >= 1

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # nothing above this line touches the GPU

    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("sus-net_amd")
    # SUSNET_BENCH_BACKEND=gloo + SUSNET_BENCH_ONE_DEVICE=1 rehearse the N > 1 control flow on a 1-GPU box
    backend = os.environ.get("SUSNET_BENCH_BACKEND", "nccl")
    if os.environ.get("SUSNET_BENCH_ONE_DEVICE") == "1":
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = pkg.dist.init_from_env(backend)
    check_world(world, args.gpus)
    if world > 1:  # every rank must be there: the gathered world is what n_gpus reports
        seen = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local) if backend == "nccl" else "cpu")
        dist.all_reduce(seen)
        assert int(seen.item()) == args.gpus, f"gathered {int(seen.item())} ranks, expected {args.gpus}"
    if args.dry_run:
        dry_run(args, pkg, rank, world)
        return
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    coll_dev = device if backend == "nccl" else "cpu"
    seed = 1234

    def obs_config(mode):
        if mode == "none":
            return None
        if mode == "raw":
            return pkg.ObsConfig("raw", dtype=torch.uint8)
        if mode == "flat":
            return pkg.ObsConfig("flat", ["onehot_pos"], dtype=torch.float32)
        return pkg.ObsConfig("planes", dtype=torch.float32)

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def run_fused(env, n_launches, T, bufs, events=None):
        stream = torch.cuda.current_stream(device)
        for _ in range(n_launches):
            if events is not None:  # HIP events bracketing THIS launch on the launch stream
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                env.rollout_into(T, bufs)
                e1.record(stream)
                events.append((e0, e1))
            else:
                env.rollout_into(T, bufs)
        return n_launches

    def run_step(env, n_ticks):
        for _ in range(n_ticks):
            a = env.sample_actions()
            env.step(a)
        return 2 * n_ticks

    def measure(spec, B, mode, obs_mode, K, W, ticks, want_packed=-1, graph_ticks=0, policy_fused=True, n_repeats=0, block_ticks=0, crew_network=False):
        """W untimed + K timed bench steps of one configuration.  fused: a step is one launch of `ticks` ticks; step / policy:
        one tick (policy with graph_ticks > 0: the tick loop replayed as hipGraphs of graph_ticks ticks, K rounded up to whole graphs)."""
        oc = obs_config(obs_mode) if mode == "fused" else None
        step_obs = obs_config(obs_mode) if mode == "step" else None
        if mode == "policy":
            step_obs = pkg.ObsConfig("flat", POLICY_COMPONENTS)
        env = make_env(pkg, spec, B, seed, rank * B, device, obs_cfg=step_obs)
        env.reset()
        lay = env.record_layout()
        packed = (mode == "fused" and obs_mode == "raw" and want_packed != 0 and lay is not None)
        if want_packed in (1, 2) and mode == "fused" and obs_mode == "raw":
            assert packed, "this configuration has no packed record mode"
        compact = packed and want_packed in (-1, 2) and env.record_layout("compact") is not None
        if want_packed == 2:
            assert compact, "this configuration has no compact record"
        if compact:
            lay = env.record_layout("compact")
        bufs = env.alloc_rollout(ticks, obs=oc, packed=("compact" if compact else packed)) if mode == "fused" else None
        if mode == "policy":
            model = pkg.policy.reference_imposter_mlp(env, POLICY_COMPONENTS, seed=0)
            crew_model = pkg.policy.reference_crew_mlp(env, POLICY_COMPONENTS, seed=1) if crew_network else None  # (run_game: both teams by their networks)
            pr = pkg.PolicyRollout(env, model, crew_model=crew_model, components=POLICY_COMPONENTS, fused=policy_fused)
            assert (pr.fused_imposter is not None) == policy_fused, "the reference MLP on this layout is served by susnet_qnet_forward"
            if graph_ticks > 0:
                graph, _ = pr.capture(graph_ticks)
                K = (K + graph_ticks - 1) // graph_ticks * graph_ticks
                W = (W + graph_ticks - 1) // graph_ticks * graph_ticks

                def runner(n):
                    for _ in range(n // graph_ticks):
                        graph.replay()
                    return n // graph_ticks
            else:
                def runner(n):  # (block_ticks > 0: that many ticks per launch where the env serves the whole tick as one kernel)
                    pr.run(n, block_ticks=block_ticks)
                    return n if block_ticks <= 0 else (n + block_ticks - 1) // block_ticks
        else:
            runner = (lambda n: run_fused(env, n, ticks, bufs)) if mode == "fused" else (lambda n: run_step(env, n))
        if args.settle_ms > 0:  # steady clocks first (see --settle-ms): untimed, not part of the W warm-up steps
            unit = graph_ticks if (mode == "policy" and graph_ticks > 0) else 1
            t_s = time.perf_counter()
            while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
                runner(4 * unit)
                torch.cuda.synchronize(device)
        runner(W)
        sync_all()
        stream = torch.cuda.current_stream(device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        launches = runner(K)
        ev1.record(stream)
        sync_all()
        dt = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)  # HIP events around the timed region, on the launch stream
        # the same K-launch region `repeats` more times, back to back (outside the headline's timed region): how much one such
        # measurement moves on this box
        rep = None
        if n_repeats > 0 and mode == "fused":
            vals = []
            for _ in range(n_repeats):
                r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                r0.record(stream)
                runner(K)
                r1.record(stream)
                torch.cuda.synchronize(device)
                vals.append(r0.elapsed_time(r1) * 1e3 / K)
            vals.sort()
            rep = {"n": n_repeats, "launches_each": K, "median_launch_us": vals[len(vals) // 2], "min_launch_us": vals[0], "max_launch_us": vals[-1]}
        pair_us = None
        if mode == "fused":  # cross-check OUTSIDE the timed region: 8 launches, each bracketed by its own event pair
            pairs = []
            run_fused(env, 8, ticks, bufs, pairs)
            torch.cuda.synchronize(device)
            pair_us = sum(a.elapsed_time(b) for a, b in pairs) * 1e3 / len(pairs)
        launch_us_ranks = None
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
            mine = torch.tensor([dev_ms * 1e3 / K], dtype=torch.float64, device=coll_dev)
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
            launch_us_ranks = [float(v.item()) for v in allv]
        env.poll_errors()
        metrics = pkg.dist.node_metrics(env)  # the ONE collective: all-gather of the episode totals
        qnet = None
        if mode == "policy" and policy_fused:  # the network kernel by itself, outside the timed region: 50 launches between two events
            net = pr.fused_imposter
            for _ in range(5):
                env.qnet_forward(net)
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            q0.record(stream)
            for _ in range(50):
                env.qnet_forward(net)
            q1.record(stream)
            torch.cuda.synchronize(device)
            us = q0.elapsed_time(q1) * 1e3 / 50
            pad = [net.dims[0], 256, 128, 64, 32, 32]
            issued = 2.0 * B * sum(a * b for a, b in zip(pad[1:-1], pad[2:]))       # layers 2..5 at their padded widths: what the matrix core executes
            model_flops = 2.0 * B * sum(a * b for a, b in zip(net.dims[:-1], net.dims[1:]))  # the reference MLP's own multiply-adds, layer 1 included
            traffic, traffic_source = None, None
            import glob

            cpaths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_qnet_counters.json")))
            if B == 65536 and cpaths:  # WRITE_SIZE + 2 x FETCH_SIZE of the same kernel on the same batch (tools/profile_cfg5.sh): the newest round's
                traffic, traffic_source = json.load(open(cpaths[-1])).get("traffic_bytes_per_launch"), "profiles/" + os.path.basename(cpaths[-1])
            alone = {"kernel": "k_qnet<FlatRow<2,3,14>> (susnet_qnet_forward), 50 launches outside the timed region", "avg_launch_us": us,
                     "achieved": issued / us / 1e6, "frac": issued / us / 1e6 / MFMA_F32_PEAK_TFLOPS, "model_tflops": model_flops / us / 1e6,
                     "traffic": traffic, "traffic_source": traffic_source}
            one = bool(getattr(pr, "one_kernel_tick", False))
            # the kernel of the TIMED region: with the one-kernel tick every launch there is k_qnet_step (network + argmax + crew draws + step:
            # the matrix core idles while the wave steps its environments); else the network kernel as timed alone
            tick_us = dev_ms * 1e3 / K if one else us
            qnet = {"kernel": (("k_qnet_step<FlatRow<2,3,14>, Spec<3,4,..>> (susnet_qnet_policy_step: the whole tick, one launch per bench step)" if block_ticks <= 0 else
                                f"k_qnet_step<FlatRow<2,3,14>, Spec<3,4,..>> (susnet_qnet_policy_rollout: the whole tick, {block_ticks} ticks = bench steps per launch; "
                                "avg_launch_us is per TICK)") if one
                               else "k_qnet<FlatRow<2,3,14>> (susnet_qnet_forward)"),
                    "bound": "mfma", "avg_launch_us": tick_us, "unit": "TFLOP/s", "traffic": None if one else traffic,
                    "peak": MFMA_F32_PEAK_TFLOPS, "achieved": issued / tick_us / 1e6, "frac": issued / tick_us / 1e6 / MFMA_F32_PEAK_TFLOPS,
                    "model_flops_per_launch": model_flops, "model_tflops": model_flops / tick_us / 1e6, "network_kernel_alone": alone,
                    "note": "achieved = flops of the v_mfma_f32_32x32x2_f32 instructions issued (layers 2..5, padded widths) / launch time; layer 1 "
                            "(35 % of the model's multiply-adds) is a gather of W1 columns and issues no MFMA, so model_tflops exceeds it"}
        return dict(seconds=dt, launches=launches, device_ms=dev_ms, metrics=metrics, pair_us=pair_us, packed=packed, steps=K, warmup=W,
                    raw_size=env.flattened_state_size, record_bytes=lay.record_bytes if lay is not None else None,
                    launch_us_ranks=launch_us_ranks, qnet=qnet, repeats=rep,
                    layout=("compact record per env-step (16 bytes: rewards f32[2] | raw obs u8[6] | actions and flags in one byte | 0)" if mode == "fused" and compact
                            else ("packed record per env-step" + (", stored as planes of 16-byte pieces" if lay is not None and lay.planar else "")) if packed
                            else "separate tensors"))

    K, W = args.steps, args.warmup
    spec = CONFIGS[args.config]
    A = spec["A"]
    B = args.batch or spec["batch"]
    if spec.get("policy"):
        args.mode, args.obs = "policy", "flat"
    ticks_per_step = args.ticks if args.mode == "fused" else 1
    res = measure(spec, B, args.mode, args.obs, K, W, args.ticks, args.packed, n_repeats=args.repeats if world == 1 else 0,
                  block_ticks=args.policy_block if args.mode == "policy" else 0)
    steps_per_launch = B * ticks_per_step          # env-steps one launch of the dominant kernel processes (this rank)
    total_steps = steps_per_launch * world * K
    value = total_steps / res["seconds"]
    # roofline of the dominant kernel: bytes per launch / its average duration over the TIMED region (HIP events on the launch
    # stream around the K launches; fused: K back-to-back k_rollout launches and nothing else)
    avg_launch_s = (res["device_ms"] / 1e3) / K
    roof = roofline_block(args.config, spec, B, ticks_per_step, args.obs, res["packed"], res["raw_size"], res["record_bytes"], avg_launch_s,
                          res["pair_us"], with_traffic=(args.mode == "fused" and world == 1))
    if args.mode == "step":
        roof["kernel"] = "k_sample_philox + k_step<PhiloxRng, Spec>"
    if res["launch_us_ranks"]:
        roof["avg_launch_us_per_rank"] = {"min": min(res["launch_us_ranks"]), "max": max(res["launch_us_ranks"]), "all": res["launch_us_ranks"]}
    line = {
        "metric": "env-steps/s", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": res["seconds"] * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic", "settle_ms": args.settle_ms,
        "config": config_block(spec, args.mode, args.obs, B, world, ticks_per_step, res["layout"], value * A),
        "roofline": roof,
        "episode_metrics": {k: v for k, v in res["metrics"].items() if k != "per_rank_episodes"},
    }
    if res["repeats"]:
        rp = res["repeats"]
        rp["median_value"] = steps_per_launch / (rp["median_launch_us"] * 1e-6)
        rp["note"] = "the K-launch timed region repeated back to back after the headline measurement (HIP events per repeat); value at the median launch time"
        line["repeats"] = rp
    if args.mode == "policy":  # the tick's dominant kernel is the Q-network: its roofline is the f32 matrix peak
        line["roofline"] = dict(res["qnet"])
        line["dtype"] = "f32"
        line["config"]["policy_forward"] = ("the whole tick -- Q-network (float32, f32-input MFMA), argmax, crew draws, env step -- as one HIP kernel; "
                                            + (f"{args.policy_block} ticks per launch (susnet_qnet_policy_rollout)" if args.policy_block > 0 else
                                               "one launch per tick (susnet_qnet_policy_step)"))
        line["config"]["ticks_per_launch"] = max(1, args.policy_block)
        line["headline_definition"] = (f"cfg5 value = batch x ticks / wall time of the policy loop at {max(1, args.policy_block)} tick(s) per launch with fixed weights "
                                       "(default 5 = the reference trainer's train_step_interval: rounds 1-3 timed one launch per tick, round 4 blocks of 64 ticks -- "
                                       "both are in `policy_forms`)")
        forms = {}
        for label, blk in (("one_launch_per_tick", 0), ("blocks_of_5_ticks", 5), ("blocks_of_64_ticks", 64)):
            if blk == args.policy_block:
                forms[label] = {"value": value, "us_per_tick": res["seconds"] * 1e6 / K, "ticks_per_launch": max(1, blk), "headline": True}
                continue
            rp = measure(spec, B, "policy", "flat", 320 if blk else 64, 64 if blk else 16, 1, block_ticks=blk)
            forms[label] = {"value": B * rp["steps"] / rp["seconds"], "us_per_tick": rp["seconds"] * 1e6 / rp["steps"], "ticks_per_launch": max(1, blk)}
            del rp
        line["policy_forms"] = forms
    del res
    secondary = rank == 0 and world == 1 and not args.no_secondary and args.mode != "policy"
    if secondary:
        other = "step" if args.mode == "fused" else "fused"
        k2 = 512 if other == "step" else 8
        r2 = measure(spec, B, other, args.obs, k2, 64 if other == "step" else 2, args.ticks, args.packed)
        per2 = B * (1 if other == "step" else args.ticks)
        line["secondary"] = {"mode": other, "value": per2 * k2 / r2["seconds"], "unit": "env-steps/s",
                             "ms_per_tick": r2["seconds"] * 1e3 / (k2 * (1 if other == "step" else args.ticks)),
                             "launches_per_tick": 2 if other == "step" else 1.0 / args.ticks}
        del r2
        if other == "step":
            # the same two kernels per tick captured as a hipGraph of 16 ticks and replayed (the step counter of the
            # action stream lives in device memory, so a replay draws fresh actions)
            env_g = make_env(pkg, spec, B, seed, rank * B, device, obs_cfg=obs_config(args.obs))
            env_g.reset()
            gt = 16
            graph = env_g.capture_random_step(gt)
            for _ in range(8):
                graph.replay()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(4096 // gt):
                graph.replay()
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            line["secondary"]["hip_graph_replay"] = {"value": B * 4096 / dt, "unit": "env-steps/s", "ms_per_tick": dt * 1e3 / 4096,
                                                     "ticks_per_graph": gt, "nodes_per_tick": env_g.graph_nodes_per_tick()}
            del env_g, graph
        # the same rollout with the reference's float32 feature layouts fused in (HBM-write bound)
        line["obs_modes"] = []
        for om in ("flat", "planes"):
            if om == args.obs:
                continue
            k3, t3 = 4, min(args.ticks, 128)  # planes f32 at 128 ticks is already an 11 GB trajectory
            r3 = measure(spec, B, "fused", om, k3, 1, t3)
            per_launch_s = (r3["device_ms"] / 1e3) / k3
            rb = roofline_block(args.config, spec, B, t3, om, False, r3["raw_size"], None, per_launch_s, with_traffic=False)
            line["obs_modes"].append({
                "obs": om + ("(onehot_pos)" if om == "flat" else "") + " f32", "value": B * t3 * k3 / r3["seconds"], "unit": "env-steps/s",
                "bytes_per_env_step": rb["bytes_per_env_step"], "achieved_GBs": rb["achieved"], "frac": rb["frac"],
                "survey_8d_bytes_per_env_step": rb["survey_8d_bytes_per_env_step"], "survey_8d_frac": rb["survey_8d_frac"],
                "ticks_per_launch": t3, "avg_launch_us": per_launch_s * 1e6})
            del r3
        # the other BASELINE configurations, each measured like the headline (5 warm-up + 20 timed launches of 512 ticks)
        line["other_configs"] = []
        for name in ("cfg2w", "cfg3", "cfg4", "tag5", "cfg5"):
            if name == args.config:
                continue
            sp = CONFIGS[name]
            if sp.get("policy"):
                entry = {"config": name, "workload": sp["workload"], "unit": "env-steps/s"}
                # one launch per tick (eager, and replayed as hipGraphs of 8 ticks), the torch-module network as the comparison, and the
                # loop as bench.py --config cfg5 times it: 64 ticks per launch (run_game's loop runs with fixed networks)
                for label, gt, fused, blk in (("eager", 0, True, 0), ("hip_graph_replay", 8, True, 0), ("torch_modules_hip_graph_replay", 8, False, 0),
                                              ("blocks_of_5_ticks", 0, True, 5), ("blocks_of_64_ticks", 0, True, 64)):
                    rp = measure(sp, sp["batch"], "policy", "flat", 320 if blk else 64, 64 if blk else 16, 1, graph_ticks=gt, policy_fused=fused, block_ticks=blk)
                    entry[label] = {"value": sp["batch"] * rp["steps"] / rp["seconds"], "us_per_tick": rp["seconds"] * 1e6 / rp["steps"],
                                    "ticks_timed": rp["steps"], **({"ticks_per_graph": gt} if gt else {}), **({"ticks_per_launch": blk} if blk else {})}
                    if rp["qnet"] is not None and (blk == 5 or "roofline" not in entry):  # (the roofline of the form `value` is taken from: 5 ticks per launch)
                        entry["roofline"] = rp["qnet"]
                    del rp
                # BOTH teams by their networks (visualize.py:547-562: run_game's loop): the same kernel swapping the LDS image between the two
                # network passes of a tick -- one launch per tick, and 64 ticks per launch
                for label, blk in (("both_teams_one_launch_per_tick", 0), ("both_teams_blocks_of_64_ticks", 64)):
                    rp = measure(sp, sp["batch"], "policy", "flat", 256 if blk else 64, 64 if blk else 16, 1, block_ticks=blk, crew_network=True)
                    entry[label] = {"value": sp["batch"] * rp["steps"] / rp["seconds"], "us_per_tick": rp["seconds"] * 1e6 / rp["steps"], "ticks_timed": rp["steps"],
                                    "networks": "imposter MLP[88,256,128,64,16,7] + crew MLP[88,256,128,64,16,6], both inside k_qnet_step<.., TWO>"}
                    del rp
                # the trainer's collection loop on the same env: the one-kernel tick writing the replay feed + one susnet_ring_append per
                # 64 ticks (DeviceReplayBuffer.collect: train.py:345-399), epsilon-greedy as the trainer acts
                envc = make_env(pkg, sp, sp["batch"], seed, rank * sp["batch"], device, obs_cfg=pkg.ObsConfig("flat", POLICY_COMPONENTS))
                envc.reset()
                polc = pkg.PolicyRollout(envc, pkg.policy.reference_imposter_mlp(envc, POLICY_COMPONENTS, seed=0), None, components=POLICY_COMPONENTS,
                                         epsilon=0.1, mask_dead=True)
                ring = pkg.DeviceReplayBuffer(1 << 21, envc.flattened_state_size, 1, envc.n_agents, envc.n_imposters, device=device)
                ring.collect(envc, polc, 64, epsilon=0.1)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                n_added = ring.collect(envc, polc, 256, epsilon=0.1)
                torch.cuda.synchronize(device)
                dtc = time.perf_counter() - t0
                entry["collect_into_replay_ring"] = {"value": n_added / dtc, "unit": "transitions/s", "ticks": 256, "ticks_per_append": 64, "epsilon": 0.1,
                                                     "us_per_tick": dtc * 1e6 / 256,
                                                     "note": "DeviceReplayBuffer.collect: susnet_qnet_policy_rollout (network, epsilon-greedy argmax, random crew, step, "
                                                             "replay feed: one launch per 64-tick block) + susnet_ring_append per block; the reference's six ring tensors, "
                                                             "trajectory_size 1"}
                # the same tick looped inside ONE launch per 64-tick block (susnet_qnet_policy_rollout: the network image is loaded and
                # the launch paid once per block; what collect() above uses)
                if envc.supports_qnet_policy_step(polc.fused_imposter):
                    feed = envc.alloc_feed(64)
                    for _ in range(2):
                        envc.policy_rollout_into(feed, 64, polc.fused_imposter, epsilon=0.1, mask_dead=True)
                    b0e, b1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    b0e.record(torch.cuda.current_stream(device))
                    for _ in range(4):
                        envc.policy_rollout_into(feed, 64, polc.fused_imposter, epsilon=0.1, mask_dead=True)
                    b1e.record(torch.cuda.current_stream(device))
                    torch.cuda.synchronize(device)
                    us_tick = b0e.elapsed_time(b1e) * 1e3 / (4 * 64)
                    entry["block_with_replay_feed"] = {"value": sp["batch"] / (us_tick * 1e-6), "unit": "env-steps/s", "us_per_tick": us_tick, "ticks_per_launch": 64,
                                                    "launches_timed": 4, "epsilon": 0.1,
                                                    "note": "susnet_qnet_policy_rollout: k_qnet_step looping over 64 ticks per launch, replay feed written"}
                    del feed
                del envc, polc, ring
                # the loop as the reference trainer runs it between two optimizer steps: 5 ticks with fixed weights (train_step_interval, train.py:295)
                entry["value"], entry["value_from"] = entry["blocks_of_5_ticks"]["value"], "blocks_of_5_ticks"
                entry["kernel"] = ("k_qnet_step (float32 Q-network on the f32-input MFMA, argmax, crew draws and the env step as one kernel: one launch per tick -- "
                                   "susnet_qnet_policy_step -- or 64 ticks per launch -- susnet_qnet_policy_rollout)")
                entry["torch_modules_hip_graph_replay"]["note"] = ("the same tick with the network as torch modules (hipBLASLt f32 GEMMs + PReLU kernels on "
                                                                   "the [B][88] observation): round 2's path, kept as the comparison")
                line["other_configs"].append(entry)
                continue
            ro = measure(sp, sp["batch"], "fused", "raw", 20, 5, 512)
            launch_s = (ro["device_ms"] / 1e3) / 20
            rb = roofline_block(name, sp, sp["batch"], 512, "raw", ro["packed"], ro["raw_size"], ro["record_bytes"], launch_s, ro["pair_us"])
            line["other_configs"].append({
                "config": name, "workload": sp["workload"], "value": sp["batch"] * 512 * 20 / ro["seconds"], "unit": "env-steps/s",
                "steps": 20, "warmup": 5, "ticks_per_launch": 512, "batch": sp["batch"],
                "trajectory_layout": ro["layout"],
                "avg_launch_us": rb["avg_launch_us"], "avg_launch_us_event_pairs": rb["avg_launch_us_event_pairs"], "kernel": rb["kernel"],
                "bytes_per_env_step": rb["bytes_per_env_step"], "achieved_GBs": rb["achieved"], "frac": rb["frac"],
                "survey_8d_bytes_per_env_step": rb["survey_8d_bytes_per_env_step"], "survey_8d_frac": rb["survey_8d_frac"],
                "traffic": rb["traffic"], "traffic_frac": rb["traffic_frac"], "traffic_source": rb["traffic_source"],
                "episodes": ro["metrics"].get("episodes"), "episode_steps": ro["metrics"].get("episode_steps")})
            del ro
    if "other_configs" in line:  # the same numbers in a compact top-level key (the long entries above get truncated in driver records)
        line["others"] = {e["config"]: {"value": e["value"], "frac": (e.get("frac") if "frac" in e else e.get("roofline", {}).get("frac")),
                                        **({"collect_transitions_per_s": e["collect_into_replay_ring"]["value"]} if "collect_into_replay_ring" in e else {}),
                                        **({"us_per_tick": e["blocks_of_5_ticks"]["us_per_tick"], "us_per_tick_blocks_of_64": e["blocks_of_64_ticks"]["us_per_tick"],
                                            "launch_per_tick_us": e["eager"]["us_per_tick"],
                                            "both_teams_us_per_tick": e.get("both_teams_one_launch_per_tick", {}).get("us_per_tick"),
                                            "both_teams_us_per_tick_blocks_of_64": e.get("both_teams_blocks_of_64_ticks", {}).get("us_per_tick")}
                                           if "blocks_of_64_ticks" in e else {})}
                          for e in line["other_configs"]}
    if world > 1:
        dist.barrier()  # the timed region and its collectives are over on every rank before rank 0 spends host time below
    if rank == 0 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(spec)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
