"""CPU-only properties of the oracle's word sources and of the production (Philox) draw protocol."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_sanitizer_selftest():
    """AddressSanitizer + UBSan walk over every env class / word source / observation (CPU build only)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "selftest"])
    out = subprocess.check_output([os.path.join(ROOT, "oracle", "_build", "selftest")], text=True)
    assert "selftest ok" in out


def _run(ob, steps):
    ob.reset()
    log = []
    for _ in range(steps):
        a = ob.sample_actions()
        rew, done, trunc, rc = ob.step(a)
        assert rc == 0
        log.append((a.copy(), rew.copy(), done.copy(), trunc.copy()))
        ob.reset(mask=(done | trunc).astype(bool))
    return log, ob.export()


def test_tape_of_mt19937_words_equals_mt19937_mode(oracle_mod):
    """Feeding numpy's raw stream as a tape reproduces the seeded run: the TAPE source has numpy semantics."""
    om = oracle_mod
    B, steps = 6, 300
    for variant, kw in (("base", dict(n_imposters=2, n_crew=5, n_jobs=3)), ("tagging", dict(n_imposters=1, n_crew=3, n_jobs=2, tag_reset_interval=5)),
                        ("itg", dict(n_crew=3, n_jobs=1, shuffle_imposter_index=True))):
        cfg = om.make_config(variant, max_time_steps=40, **kw)
        a = om.OracleBatch(cfg, B)
        a.seed_mt(range(100, 100 + B))
        b = om.OracleBatch(cfg, B)
        b.set_tapes(np.stack([om.mt19937_words(100 + i, 20000) for i in range(B)]))
        la, ea = _run(a, steps)
        lb, eb = _run(b, steps)
        for x, y in zip(la, lb):
            for u, v in zip(x, y):
                np.testing.assert_array_equal(u, v)
        for k in ea:
            np.testing.assert_array_equal(ea[k], eb[k], err_msg=k)
        assert not b.tape_overflow.any()


def _chi2(counts, expected):
    return float(((counts - expected) ** 2 / expected).sum())


def test_philox_protocol_draws_are_uniform(oracle_mod):
    """The production protocol claims the reference's DISTRIBUTIONS (uniform spawn cells with replacement,
    uniform job cells without replacement, uniform role-valid actions, uniform imposter index): chi-square."""
    om = oracle_mod
    B = 20000
    cfg = om.make_config("base", n_imposters=1, n_crew=3, n_jobs=4)
    ob = om.OracleBatch(cfg, B)
    ob.set_philox(2024, 0, 0)
    ob.reset()
    e = ob.export()
    grid = om.reference_grid(True)
    valid = np.argwhere(grid)
    nv = len(valid)  # 68 free cells
    cell_id = {tuple(v): i for i, v in enumerate(valid)}
    # spawn cells: uniform over the 68 free cells (df = 67; 99.9 % quantile ~ 110)
    ids = np.array([cell_id[tuple(p)] for p in e["pos"].reshape(-1, 2)])
    c = np.bincount(ids, minlength=nv).astype(float)
    assert _chi2(c, len(ids) / nv) < 120
    # job cells: distinct within an env, uniform marginally
    jid = np.array([[cell_id[tuple(p)] for p in row] for row in e["jobpos"]])
    assert (np.sort(jid, axis=1)[:, 1:] != np.sort(jid, axis=1)[:, :-1]).all()
    c = np.bincount(jid.ravel(), minlength=nv).astype(float)
    assert _chi2(c, jid.size / nv) < 120
    # imposter index uniform over the 4 agents (df = 3; 99.9 % ~ 16.3)
    imp = e["imp"].argmax(1)
    assert _chi2(np.bincount(imp, minlength=4).astype(float), B / 4) < 17
    # actions: crew uniform over 6, imposter over 7
    a = ob.sample_actions()
    is_imp = e["imp"].astype(bool)
    ci = np.bincount(a[is_imp], minlength=7).astype(float)
    cc = np.bincount(a[~is_imp], minlength=6).astype(float)
    assert a[~is_imp].max() == 5 and a[is_imp].max() == 6
    assert _chi2(ci, ci.sum() / 7) < 23 and _chi2(cc, cc.sum() / 6) < 21
    # streams of different envs / seeds differ, same (seed, env id) repeats
    ob2 = om.OracleBatch(cfg, 4)
    ob2.set_philox(2024, 0, 0)
    ob2.reset()
    np.testing.assert_array_equal(ob2.export()["pos"], e["pos"][:4])
    ob2.set_philox(2025, 0, 0)
    ob2.reset()
    assert not np.array_equal(ob2.export()["pos"], e["pos"][:4])


def test_one_word_serves_both_1v1_agents_jointly_uniform(oracle_mod):
    """1v1 production protocol: ONE action-stream word per tick gives both agents' actions by nested multiply-shift.
    The pair (imposter action in 0..5, crew action in 0..4) must be jointly uniform over its 30 cells, and consecutive
    ticks independent (df = 29: 99.9 % quantile ~ 58; df = 35 for the lag pairs: ~ 66)."""
    om = oracle_mod
    B, ticks = 3000, 40
    cfg = om.make_config("itg", n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                         time_step_reward=0, include_walls=False, shuffle_imposter_index=False)
    ob = om.OracleBatch(cfg, B)
    ob.set_philox(77, 0, 0)
    ob.reset()
    joint = np.zeros((6, 5))
    lag = np.zeros((6, 6))
    prev = None
    for t in range(ticks):
        a = ob.sample_actions()
        assert a[:, 0].max() == 5 and a[:, 1].max() == 4 and a.min() == 0
        np.add.at(joint, (a[:, 0], a[:, 1]), 1)
        if prev is not None:
            np.add.at(lag, (prev[:, 0], a[:, 0]), 1)
        prev = a
        _, done, trunc, _ = ob.step(a)
        ob.reset(mask=(done | trunc).astype(bool))
    assert _chi2(joint.ravel(), B * ticks / 30) < 58
    assert _chi2(lag.ravel(), B * (ticks - 1) / 36) < 66
    # sampling twice before a step returns the same actions (the step advances the tick, sampling does not)
    np.testing.assert_array_equal(ob.sample_actions(), ob.sample_actions())


def test_action_order_from_one_word_is_a_uniform_permutation(oracle_mod):
    """Production protocol: the step's shuffled agent order is decoded from ONE action-stream word (Fisher-Yates digits by
    nested multiply-shift).  All 3! orders of a 1v2 game must be equally likely (df = 5: 99.9 % quantile ~ 20.5), every
    agent of a 2v6 game equally likely at every turn (8 x 8 table, df = 49: ~ 85), and consecutive steps unrelated."""
    om = oracle_mod
    B = 12000
    ob = om.OracleBatch(om.make_config("base", n_imposters=1, n_crew=2, n_jobs=4), B)
    ob.set_philox(5, 0, 0)
    ob.reset()
    counts, lag = {}, np.zeros((3, 3))
    prev = None
    for _ in range(4):
        ob.step(np.zeros((B, 3), dtype=np.int64))
        order = ob.order.copy()
        assert (np.sort(order, axis=1) == np.arange(3)).all()
        for row in map(tuple, order):
            counts[row] = counts.get(row, 0) + 1
        if prev is not None:
            np.add.at(lag, (prev[:, 0], order[:, 0]), 1)
        prev = order
    assert len(counts) == 6 and _chi2(np.array(list(counts.values()), dtype=float), 4 * B / 6) < 20.5
    assert _chi2(lag.ravel(), 3 * B / 9) < 26.1  # df = 8
    ob8 = om.OracleBatch(om.make_config("base", n_imposters=2, n_crew=6, n_jobs=4), B)
    ob8.set_philox(6, 0, 0)
    ob8.reset()
    ob8.step(np.zeros((B, 8), dtype=np.int64))
    o8 = ob8.order
    assert (np.sort(o8, axis=1) == np.arange(8)).all()
    table = np.zeros((8, 8))
    for turn in range(8):
        table[turn] = np.bincount(o8[:, turn], minlength=8)
    assert _chi2(table.ravel(), B / 8) < 85


def test_four_agents_share_one_action_word_independently(oracle_mod):
    """Production protocol, more than four action-stream words per tick: agents are packed four per word by nested
    multiply-shift.  Marginals stay uniform over each role's range and agents sharing a word are independent (a crew-crew
    pair: 6 x 6 table, df = 35: 99.9 % quantile ~ 66.6; agent 3 vs agent 4 sit in different words: same bound)."""
    om = oracle_mod
    B = 40000
    ob = om.OracleBatch(om.make_config("base", n_imposters=2, n_crew=6, n_jobs=4, shuffle_imposter_index=False), B)
    ob.set_philox(11, 0, 0)
    ob.reset()
    a = ob.sample_actions()  # agents 0, 1 imposters (7 actions), 2..7 crew (6 actions)
    assert a[:, :2].max() == 6 and a[:, 2:].max() == 5 and a.min() == 0
    for i in range(8):
        n = 7 if i < 2 else 6
        assert _chi2(np.bincount(a[:, i], minlength=n).astype(float), B / n) < (22.5 if n == 7 else 20.5), i
    for i, j in ((2, 3), (3, 4), (0, 1), (6, 7)):
        ni, nj = (7 if i < 2 else 6), (7 if j < 2 else 6)
        t = np.zeros((ni, nj))
        np.add.at(t, (a[:, i], a[:, j]), 1)
        assert _chi2(t.ravel(), B / (ni * nj)) < 85, (i, j)


def test_three_1v1_ticks_share_one_action_word_independently(oracle_mod):
    """1v1 production protocol: ONE action-stream word serves THREE consecutive ticks (six nested digits, 30^3 = 27000 << 2^32).
    The imposter's actions at ticks 3k, 3k+1, 3k+2 -- all digits of the same word -- must be jointly uniform over their
    6^3 = 216 cells (df = 215: 99.9 % quantile ~ 284), and so must (crew at 3k, imposter at 3k+2)."""
    om = oracle_mod
    B = 30000
    cfg = om.make_config("itg", n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                         time_step_reward=0, include_walls=False, shuffle_imposter_index=False, max_time_steps=100000)
    ob = om.OracleBatch(cfg, B)
    ob.set_philox(99, 0, 0)
    ob.reset()
    acts = []
    for t in range(6):
        a = ob.sample_actions()
        acts.append(a.copy())
        ob.step(np.zeros((B, 2), dtype=np.int64))  # (STAY, STAY): nobody dies, roles stay put
    for k in (0, 3):
        tri = np.zeros((6, 6, 6))
        np.add.at(tri, (acts[k][:, 0], acts[k + 1][:, 0], acts[k + 2][:, 0]), 1)
        assert _chi2(tri.ravel(), B / 216) < 284
        pair = np.zeros((5, 6))
        np.add.at(pair, (acts[k][:, 1], acts[k + 2][:, 0]), 1)
        assert _chi2(pair.ravel(), B / 30) < 58
    # a different word serves ticks 3..5: no relation to ticks 0..2
    cross = np.zeros((6, 6))
    np.add.at(cross, (acts[2][:, 0], acts[3][:, 0]), 1)
    assert _chi2(cross.ravel(), B / 36) < 66


def test_large_games_keep_every_word_below_the_bias_cap(oracle_mod):
    """The word packing closes a word before the product of its draws' largest ranges passes 2^16, whatever the agent count:
    a 4v8 game's 12! orders cannot come from one 32-bit word.  Checked on the layout itself and on the resulting order
    distribution of a 12-agent game (every agent equally likely at every turn: 12 x 12 table, df = 121: 99.9 % ~ 175;
    first-two-turns pairs: 132 cells, df = 131: ~ 187)."""
    om = oracle_mod
    B = 30000
    cfg = om.make_config("base", n_imposters=4, n_crew=8, n_jobs=2)
    ob = om.OracleBatch(cfg, B)
    env0 = ob.envs[0]
    A = 12
    words = list(env0.aw_word[:2 * A - 1])
    ranges = [7] * A + list(range(2, A + 1))  # action draws, then the placement draws k = 1 .. A-1 (range k + 1)
    assert words == sorted(words) and words[0] == 0 and env0.aw_W == words[-1] + 1 and env0.aw_tpw == 1
    for k in range(env0.aw_W):
        prod = 1
        for w, r in zip(words, ranges):
            if w == k:
                prod *= r
        assert 1 < prod <= 65536, (k, prod)
    ob.set_philox(17, 0, 0)
    ob.reset()
    ob.step(np.zeros((B, A), dtype=np.int64))
    order = ob.order
    assert (np.sort(order, axis=1) == np.arange(A)).all()
    table = np.zeros((A, A))
    for turn in range(A):
        table[turn] = np.bincount(order[:, turn], minlength=A)
    assert _chi2(table.ravel(), B / A) < 175
    first2 = np.zeros((A, A))
    np.add.at(first2, (order[:, 0], order[:, 1]), 1)
    off = first2[~np.eye(A, dtype=bool)]
    assert first2.trace() == 0 and _chi2(off, B / (A * (A - 1))) < 187
