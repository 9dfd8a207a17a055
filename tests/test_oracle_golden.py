"""The CPU oracle replays every golden trace of the reference from the recorded numpy seed alone.

This is the pin of the oracle (the reference has no tests of its own): reset, sample_actions
and step of oracle/susnet_oracle.c must reproduce the reference's state, rewards (bit pattern, -0.0
included), done, truncated, the 13 info counters, the agent order and the exact number of raw MT19937
words numpy consumed -- step by step.
"""
import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden, trace_names

NAMES = trace_names()


def _check_state(ob, g, s, tag):
    np.testing.assert_array_equal(ob.pos[0], g["pos"][s], err_msg=f"{tag} pos")
    np.testing.assert_array_equal(ob.alive[0], g["alive"][s], err_msg=f"{tag} alive")
    if ob.J:
        np.testing.assert_array_equal(ob.jobdone[0], g["jobdone"][s], err_msg=f"{tag} jobdone")
    if "used" in g:
        np.testing.assert_array_equal(ob.used[0], g["used"][s], err_msg=f"{tag} used")
        np.testing.assert_array_equal(ob.counts[0], g["counts"][s], err_msg=f"{tag} counts")
        assert ob.envs[0].cfg.tag_reset_interval - ob.timer[0] == g["timer_left"][s], f"{tag} timer"


def _check_pre(ob, g, s, tag):
    np.testing.assert_array_equal(ob.pos[0], g["pre_pos"][s], err_msg=f"{tag} pre_pos")
    np.testing.assert_array_equal(ob.alive[0], g["pre_alive"][s], err_msg=f"{tag} pre_alive")
    np.testing.assert_array_equal(ob.imp_mask[0], g["pre_imp"][s], err_msg=f"{tag} pre_imp")
    assert ob.t[0] == g["pre_t"][s], f"{tag} pre_t"
    if ob.J:
        np.testing.assert_array_equal(ob.jobpos[0], g["pre_jobpos"][s], err_msg=f"{tag} pre_jobpos")
        np.testing.assert_array_equal(ob.jobdone[0], g["pre_jobdone"][s], err_msg=f"{tag} pre_jobdone")
    if "pre_used" in g:
        np.testing.assert_array_equal(ob.used[0], g["pre_used"][s], err_msg=f"{tag} pre_used")
        np.testing.assert_array_equal(ob.counts[0], g["pre_counts"][s], err_msg=f"{tag} pre_counts")
        assert ob.timer[0] == g["pre_timer"][s]


def apply_injection(ob, g, s, b=0):
    kw = dict(pos=g["pre_pos"][s], alive=g["pre_alive"][s], t=g["pre_t"][s])
    if g["pre_jobpos"].shape[1]:
        kw.update(jobpos=g["pre_jobpos"][s], jobdone=g["pre_jobdone"][s])
    if "pre_used" in g:
        kw.update(used=g["pre_used"][s], counts=g["pre_counts"][s], timer=g["pre_timer"][s])
    ob.set_state(b, **kw)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_replays_reference_trace(oracle_mod, name):
    g = load_golden(f"{GOLDEN_DIR}/{name}.npz")
    meta = g["meta"]
    ob = oracle_mod.OracleBatch(oracle_mod.config_from_fixture_meta(meta), 1)
    assert ob.A == meta["n_agents"] and ob.J == meta["n_jobs"]
    ob.seed_mt([meta["seed"]])
    ob.reset()
    assert ob.cursor[0] == g["words"][0], "raw words consumed by the first reset"
    sampled = meta["mode"] == "sampled"
    S = len(g["done"])
    for s in range(S):
        tag = f"{name} step {s}"
        if s > 0 and g["ep_start"][s]:
            ob.reset()
        if not sampled and g["inject"][s]:
            apply_injection(ob, g, s)
        _check_pre(ob, g, s, tag)
        if sampled:
            a = ob.sample_actions()[0]
            np.testing.assert_array_equal(a, g["actions"][s], err_msg=f"{tag} sampled actions")
        else:
            a = g["actions"][s]
        rew, done, trunc, rc = ob.step_one(0, a)
        assert rc == 0
        np.testing.assert_array_equal(ob.order[0], g["order"][s], err_msg=f"{tag} order")
        _check_state(ob, g, s, tag)
        # bit-exact rewards: compare the IEEE bit patterns so -0.0 != +0.0
        assert rew.view(np.uint64).tolist() == g["rewards"][s].view(np.uint64).tolist(), f"{tag} rewards {rew} vs {g['rewards'][s]}"
        assert done == bool(g["done"][s]) and trunc == bool(g["trunc"][s]), f"{tag} done/trunc"
        np.testing.assert_array_equal(ob.metrics[0], g["metrics"][s], err_msg=f"{tag} metrics")
        nxt_reset = (s + 1 < S and g["ep_start"][s + 1])
        if s + 1 == S and (done or trunc) and ob.cursor[0] != g["words"][s + 1]:
            ob.reset()  # the generator reset once more after a final terminal step
        if not nxt_reset:
            assert ob.cursor[0] == g["words"][s + 1], f"{tag} raw words consumed"


def test_mt19937_matches_numpy_legacy_stream(oracle_mod):
    """Raw words of the restated MT19937 == numpy's legacy RandomState (numpy is on the GPU box too)."""
    for seed in (0, 1, 12345, 2**32 - 1):
        mine = oracle_mod.mt19937_words(seed, 2000)
        rs = np.random.RandomState(seed)
        theirs = rs.randint(0, 2**32, size=2000, dtype=np.uint32)  # one raw word each
        np.testing.assert_array_equal(mine, theirs)


def test_philox_known_answer(oracle_mod):
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors) + the (seed, env, cursor) mapping."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        assert tuple(oracle_mod.philox4x32_10(ctr, key)) == want
    # word d of env e under seed s = block(ctr=(d>>2 lo, d>>2 hi, e lo, e hi), key=(s lo, s hi))[d & 3]
    ob = oracle_mod.OracleBatch(oracle_mod.make_config("itg", n_crew=1), 3)
    seed, base, cur = 0x1234567890ABCDEF, (1 << 33) + 5, 4 * ((1 << 32) + 7) + 2
    ob.set_philox(seed, base, cur)
    words = ob.raw_words(9)
    for b in range(3):
        for k in range(9):
            d, e = cur + k, base + b
            blk = d >> 2
            want = oracle_mod.philox4x32_10((blk & 0xFFFFFFFF, blk >> 32, e & 0xFFFFFFFF, e >> 32),
                                            (seed & 0xFFFFFFFF, seed >> 32))[d & 3]
            assert words[b, k] == want


def _collect_names():
    import glob
    import os

    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "collect_*.npz")))


@pytest.mark.parametrize("name", _collect_names())
def test_oracle_replays_the_reference_collection_ring(oracle_mod, name):
    """tests/golden/collect_*.npz: the ring the reference's ReplayBuffer holds after the trainer's collection loop (train.py:316-322,
    345-399, 419-449 around the unmodified env / featurizer / MLPs).  The oracle, from the numpy seed and the recorded actions alone, must
    rebuild every surviving row: the pre-step state, the post-step state -- the TERMINAL one where the episode ended (`done` rows: the
    *_kills* fixtures, whose imposter network chases and kills) --, rewards, done, the acting episode's imposters; then a reset where the
    reference resets (done or truncated)."""
    g = load_golden(f"{GOLDEN_DIR}/{name}.npz")
    meta = g["meta"]
    assert meta["trajectory_size"] == 1
    ob = oracle_mod.OracleBatch(oracle_mod.config_from_fixture_meta(meta), 1)
    ob.seed_mt([meta["seed"]])
    ob.reset()
    n_steps, max_size = meta["num_steps"], meta["max_size"]
    first_kept = max(0, n_steps - max_size)
    ended = 0
    for s in range(n_steps):
        row = s % max_size
        before = ob.obs_raw_u8()[0].astype(np.int16)
        imps = list(ob.imp_idxs(0))
        rew, done, trunc, rc = ob.step(g["taken"][s][None].astype(np.int32))
        assert rc == 0
        after = ob.obs_raw_u8()[0].astype(np.int16)
        if s >= first_kept:
            np.testing.assert_array_equal(g["states"][row, 0], before, err_msg=f"{name} states row {row} (step {s})")
            np.testing.assert_array_equal(g["next_states"][row, 0], after, err_msg=f"{name} next_states row {row} (step {s})")
            np.testing.assert_array_equal(g["actions"][row], g["taken"][s], err_msg=f"{name} actions row {row}")
            assert g["rewards"][row].astype(np.float64).view(np.uint64).tolist() == rew[0].view(np.uint64).tolist(), f"{name} rewards row {row}"
            assert bool(g["dones"][row]) == bool(done[0]), f"{name} done row {row}"
            assert sorted(int(i) for i in g["imposters"][row]) == sorted(imps), f"{name} imposters row {row}"
            ended += int(done[0])
        if done[0] or trunc[0]:
            ob.reset()
    assert ended == int(g["dones"].sum())
    if "kills" in name:
        assert ended >= 15
