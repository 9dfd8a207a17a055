"""CPU-only tests of the Python host side (no kernels): grids, names, sharding arithmetic."""
import importlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("sus-net_amd")


def test_grids_match_the_reference_and_the_golden_traces(pkg):
    g9 = load_golden(f"{GOLDEN_DIR}/base_1v2_j4_s0.npz")["meta"]["grid_used"]
    np.testing.assert_array_equal(pkg.four_room_grid(9).astype(int), np.array(g9))  # base.py:171-197
    g9n = load_golden(f"{GOLDEN_DIR}/itg_1v1_nowalls_s0.npz")["meta"]["grid_used"]
    np.testing.assert_array_equal(pkg.four_room_grid(9, include_walls=False).astype(int), np.array(g9n))
    g14 = load_golden(f"{GOLDEN_DIR}/base14_1v2_j4_s0.npz")["meta"]["grid_used"]
    np.testing.assert_array_equal(pkg.four_room_grid(14).astype(int), np.array(g14))
    for n in (9, 11, 14, 16):
        g = pkg.four_room_grid(n)
        np.testing.assert_array_equal(g, g.T)  # transpose-symmetric: the grid[y, x] quirk is benign on it


def test_metric_names_and_order_are_the_reference_ones(pkg):
    want = load_golden(f"{GOLDEN_DIR}/base_1v2_j4_s0.npz")["meta"]["metric_order"]
    assert [m.value for m in pkg.SusMetrics] == want  # src/metrics.py:7-20
    assert len(want) == pkg._lib.N_METRICS


def test_action_tables_match_the_reference(pkg):
    env = pkg.env
    assert [a.value for a in env.CREW_ACTIONS] == [0, 1, 2, 3, 4, 6]          # base.py:82-89
    assert [a.value for a in env.IMPOSTER_ACTIONS] == [0, 1, 2, 3, 4, 7, 5]   # base.py:91-99
    assert [a.value for a in env.CREW_ACTIONS_SIMPLE] == [0, 1, 2, 3, 4]      # pred_prey.py:4-10
    assert [a.value for a in env.IMPOSTER_ACTIONS_SIMPLE] == [0, 1, 2, 3, 4, 5]  # pred_prey.py:12-19
    assert env.Action.KILL.is_job_action and env.Action.STAY.is_move_action


def test_shard_ranges_partition_the_batch(pkg):
    for total, world in ((262144, 8), (65536, 1), (1000, 3), (17, 16)):
        spans = [pkg.dist.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert pkg.dist.shard_range(262144, 3, 8) == (3 * 32768, 32768)  # BASELINE config 4


def test_obs_config_validation(pkg):
    with pytest.raises(AssertionError):
        pkg.ObsConfig("flat", ["nope"])
    with pytest.raises(AssertionError):
        pkg.ObsConfig("bogus")
    assert pkg.ObsConfig("planes").code == pkg._lib.OBS_PLANES


def test_env_refuses_non_gpu_devices(pkg):
    with pytest.raises(ValueError):
        pkg.BatchedFourRoomEnv(1, 2, 2, batch=2, device="cpu")
    with pytest.raises(AssertionError):
        pkg.BatchedFourRoomEnv(2, 2, 2, batch=2, device="cpu")  # reference ctor assert fires first (base.py:247)


def test_mlp_checkpoint_format_matches_reference(pkg, tmp_path):
    """{"state_dict", "config"} with the reference's module names (src/models/dqn.py:72-103, 322-329)."""
    import torch

    m = pkg.MLP([36, 256, 128, 64, 16, 6])
    keys = list(m.state_dict().keys())
    assert keys[:3] == ["model.0.weight", "model.0.bias", "model.1.weight"]  # Linear, PReLU, ...
    assert keys[-2:] == ["model.8.weight", "model.8.bias"]  # last activation dropped
    path = tmp_path / "ckpt.pt"
    m.dump_to_checkpoint(path)
    ck = torch.load(path)
    assert set(ck) == {"state_dict", "config"} and ck["config"] == {"layer_dims": [36, 256, 128, 64, 16, 6]}
    m2 = pkg.MLP.load_from_checkpoint(path)
    x = torch.randn(5, 1, 36)
    assert torch.equal(m(torch.zeros(5, 1, 1), x), m2(torch.zeros(5, 1, 1), x))
    r = pkg.RandomEquiprobable(5)(torch.zeros(7, 1))
    assert r.shape == (7, 5) and torch.all(r.sum(1) == 1)


def test_replay_ring_equals_sequential_reference_adds(pkg):
    """add_batch == N single adds of the reference ring (src/replay_memory.py:50-73), incl. wrap-around."""
    import torch

    T, S, A, NI, MAX = 2, 5, 3, 1, 7
    buf = pkg.DeviceReplayBuffer(MAX, S, T, A, NI, device="cpu")
    buf.states.zero_()  # torch.empty like the reference; zeroed here so untouched rows compare equal
    buf.actions.zero_()
    ref = dict(states=torch.zeros(MAX, T, S), actions=torch.zeros(MAX, A, dtype=torch.long), idx=0, size=0)
    g = torch.Generator().manual_seed(0)
    for n in (3, 4, 2, 9, 1):
        st = torch.rand(n, T, S, generator=g)
        ac = torch.randint(0, 6, (n, A), generator=g)
        buf.add_batch(st, ac, torch.rand(n, A, generator=g), torch.rand(n, T, S, generator=g), torch.zeros(n, dtype=torch.bool),
                      torch.zeros(n, NI, dtype=torch.int16))
        for k in range(n):
            ref["states"][ref["idx"]] = st[k]
            ref["actions"][ref["idx"]] = ac[k]
            ref["idx"] = (ref["idx"] + 1) % MAX
            ref["size"] = min(ref["size"] + 1, MAX)
        assert (buf.idx, buf.size) == (ref["idx"], ref["size"])
        assert torch.equal(buf.states, ref["states"]) and torch.equal(buf.actions, ref["actions"])
    b = buf.sample(11)
    assert b.states.shape == (11, T, S) and b.dones.shape == (11, 1) and b.imposters.dtype == torch.int16


# ------------------------------------------------------------------------------------------------
# Q-network mirrors vs the reference modules' own parameters / forward pass (tests/golden/model_*.npz)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["spatialdqn", "mlp"])
def test_q_network_mirrors_load_reference_parameters_and_reproduce_forward(name):
    import importlib
    import json

    import torch

    policy = importlib.import_module("sus-net_amd.policy")
    z = np.load(os.path.join(GOLDEN_DIR, f"model_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    cls = policy.SpatialDQN if name.startswith("spatialdqn") else policy.MLP
    model = cls(**meta["config"])
    sd = {k: torch.from_numpy(z["param::" + k]) for k in meta["keys"]}
    assert list(model.state_dict().keys()) == meta["keys"], "state_dict keys differ from the reference module's"
    model.load_state_dict(sd, strict=True)
    model.eval()
    with torch.no_grad():
        out = model(torch.from_numpy(z["input0"]), torch.from_numpy(z["input1"]))
    np.testing.assert_allclose(out.numpy(), z["output"], rtol=1e-5, atol=1e-6)
    # checkpoint round trip in the reference's {"state_dict", "config"} format (dqn.py:92-103, 295-306)
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.pt")
        model.dump_to_checkpoint(path)
        again = cls.load_from_checkpoint(path)
        with torch.no_grad():
            out2 = again(torch.from_numpy(z["input0"]), torch.from_numpy(z["input1"]))
        assert torch.equal(out, out2)
        copy = model.create_copy()
        assert all(torch.equal(a, b) for a, b in zip(copy.state_dict().values(), model.state_dict().values()))


def test_episodic_metric_handler_mirror(pkg, tmp_path):
    """src/metrics.py:67-95: per-episode lists, mean, JSON round trip; batched `step` with an ended-mask."""
    import torch

    M = pkg.SusMetrics
    h = pkg.EpisodicMetricHandler()
    h.step({m: 1 for m in M})
    info = {m: torch.tensor([2, 4, 6, 8]) for m in M}
    h.step(info, ended=torch.tensor([True, False, True, False]))
    assert h.metrics[M.IMP_KILLED_CREW] == [1, 2, 6] and h.compute()[M.CREW_WON] == 3.0
    h.set({M.AVG_CREW_RETURNS: [1.5, 2.5]})
    assert h.compute()[M.AVG_CREW_RETURNS] == 2.0
    p = tmp_path / "m.json"
    h.save_metrics(p)
    h2 = pkg.EpisodicMetricHandler()
    h2.load_metrics(p)
    assert h2.metrics["imp_killed_crew"] == [1, 2, 6]
    with pytest.raises(AssertionError):
        h.set({"nonsense": [1]})
