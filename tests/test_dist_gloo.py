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
