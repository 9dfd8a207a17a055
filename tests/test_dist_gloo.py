"""world_size-2 `gloo` test of the N > 1 path on CPU: contiguous sharding by global env id + the ONE
all-gather of the per-rank totals.  The CPU oracle stands in for the device (checker only, in tests): a
rank's shard with `env_id_base = start` must reproduce exactly the episodes a single unsharded run gives
those ids, so the all-gathered table sums to the unsharded totals."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GLOBAL_B, STEPS, SEED = 96, 120, 4242


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _totals(ob, steps):
    """Per-shard lifetime vector [episodes, crew_won, imposter_won, truncated, kills, fixes, sabotages, 0, 0, episode_steps, 0, 0]."""
    tot = np.zeros(12, dtype=np.int64)
    ob.reset()
    for _ in range(steps):
        rew, done, trunc, rc = ob.step(ob.sample_actions())
        assert rc == 0
        ended = (done | trunc).astype(bool)
        if ended.any():
            m = ob.export()["metrics"][ended]
            tot[0] += ended.sum()
            tot[1] += m[:, 8].sum()
            tot[2] += m[:, 7].sum()
            tot[3] += trunc[ended].sum()
            tot[4] += m[:, 0].sum()
            tot[5] += m[:, 4].sum()
            tot[6] += m[:, 3].sum()
            tot[9] += m[:, 6].sum()
            ob.reset(mask=ended)
    return tot


def _make_oracle(start, count):
    from oracle import oracle as om

    ob = om.OracleBatch(om.make_config("base", n_imposters=1, n_crew=2, n_jobs=4, max_time_steps=40), count)
    ob.set_philox(SEED, start, 0)
    return ob


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    pkg = importlib.import_module("sus-net_amd")
    r, w, _ = pkg.dist.init_from_env("gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    start, count = pkg.dist.shard_range(GLOBAL_B, rank, world)
    local = torch.from_numpy(_totals(_make_oracle(start, count), STEPS))
    table = pkg.dist.all_gather_totals(local)  # the one collective
    assert table.shape == (world, 12)
    assert torch.equal(table[rank], local)
    np.save(os.path.join(out_dir, f"table_{rank}.npy"), table.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_totals_equal_unsharded_totals(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    tables = [np.load(tmp_path / f"table_{r}.npy") for r in range(world)]
    np.testing.assert_array_equal(tables[0], tables[1])  # every rank holds the same gathered table
    whole = _totals(_make_oracle(0, GLOBAL_B), STEPS)
    np.testing.assert_array_equal(tables[0].sum(axis=0), whole)
    assert whole[0] > 0


def test_all_gather_totals_single_process_is_identity():
    pkg = importlib.import_module("sus-net_amd")
    v = torch.arange(12, dtype=torch.int64)
    assert torch.equal(pkg.dist.all_gather_totals(v), v.unsqueeze(0))


# ---- bench.py's N > 1 control flow (no GPU needed: everything up to the first device call) -----------------------------------
def _bench_module():
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_self_launch_builds_the_drivers_command():
    """`python bench.py --gpus N` started without the torch.distributed.run environment launches the N ranks itself, with the
    driver's own contract: one node, N processes, rendezvous on 127.0.0.1, its own flags passed through."""
    bench = _bench_module()
    cmd = bench.launch_command(4, 23456, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "23456"
    k = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    # a world that differs from --gpus is refused before a line can be printed
    bench.check_world(4, 4)
    with pytest.raises(SystemExit, match="refusing to report"):
        bench.check_world(2, 4)


def test_bench_refuses_a_world_that_differs_from_gpus_under_gloo(tmp_path):
    """Two real ranks (torch.distributed.run, gloo rendezvous on 127.0.0.1) started with --gpus 3: every rank stops at the
    world check -- non-zero exit, the refusal on stderr, no JSON line on stdout."""
    import subprocess
    import sys

    bench = _bench_module()
    env = dict(os.environ, SUSNET_BENCH_BACKEND="gloo", SUSNET_BENCH_ONE_DEVICE="1")
    cmd = bench.launch_command(2, _free_port(), ["--gpus", "3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0
    assert "refusing to report" in p.stderr and '{"metric"' not in p.stdout


def test_baseline_config_4_launch_shape_under_gloo():
    """BASELINE.json configs[3]: `2v6, 14x14 walled grid, batch = 262 144 sharded across 8 x MI355X`.  The exact command --
    `python bench.py --gpus 8 --config cfg4` as the driver launches it (torch.distributed.run, 8 ranks, 127.0.0.1) -- rehearsed with
    `--dry-run` under gloo: every rank starts, passes the world check, runs the real run's collectives on host tensors and owns the env
    ids [r * 32 768, (r + 1) * 32 768); rank 0's line names the whole-job batch and the data-parallel degree.  No device call is made
    (an 8-GPU node is the driver's to use; this keeps its first real run from being the first time the shape executes)."""
    import json
    import subprocess

    bench = _bench_module()
    env = dict(os.environ, SUSNET_BENCH_BACKEND="gloo")
    cmd = bench.launch_command(8, _free_port(), ["--gpus", "8", "--config", "cfg4", "--steps", "2", "--warmup", "1", "--dry-run"])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["dry_run"] is True and line["value"] is None and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["global_batch"] == 262144 and cfg["batch_per_gpu"] == 32768 and cfg["parallelism"] == "dp8"
    assert cfg["workload"].startswith("cfg4: FourRoomEnv 2v6, 14x14") and cfg["env_steps_per_bench_step"] == 262144 * 512
    assert line["shard_first_env_ids"] == [r * 32768 for r in range(8)]
