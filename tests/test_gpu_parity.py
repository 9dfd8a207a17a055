"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP kernels, called through the C ABI, must
be bit-exact against (a) the golden traces of the reference and (b) the CPU oracle."""
import importlib
import re

import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, crc_names, load_golden, step_record_bytes, trace_names

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return importlib.import_module("sus-net_amd")


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def env_from_meta(pkg, meta, batch, **kw):
    cls = meta["class"]
    k = dict(meta["kwargs"])
    k.pop("include_walls", None)
    grid = np.array(meta["grid_used"], dtype=bool)
    if cls == "itg":
        return pkg.BatchedImposterTrainingGround(**k, grid=grid, batch=batch, **kw)
    if cls == "tagging":
        return pkg.BatchedFourRoomEnvWithTagging(**k, grid=grid, batch=batch, **kw)
    return pkg.BatchedFourRoomEnv(**k, grid=grid, batch=batch, **kw)


def families():
    fam = {}
    for n in trace_names():
        m = re.match(r"(.*)_s(\d+)$", n)
        key = m.group(1) if m else n
        fam.setdefault(key, []).append(n)
    return fam


FAMILIES = families()


def np_(t):
    return t.detach().cpu().numpy()


def check_step_against_golden(env, gs, s, rew, done, trunc, tagging):
    B = len(gs)
    pos, alive = np_(env.agent_positions), np_(env.alive_agents)
    met = np_(env._metrics)
    for b, g in enumerate(gs):
        tag = f"{g['name']} step {s}"
        np.testing.assert_array_equal(pos[b], g["pos"][s], err_msg=tag + " pos")
        np.testing.assert_array_equal(alive[b], g["alive"][s].astype(bool), err_msg=tag + " alive")
        if env.n_jobs:
            np.testing.assert_array_equal(np_(env.completed_jobs)[b], g["jobdone"][s].astype(bool), err_msg=tag + " jobdone")
        want = g["rewards"][s]
        got = np_(rew)[b].astype(np.float64)
        assert got.view(np.uint64).tolist() == want.view(np.uint64).tolist(), f"{tag} rewards {got} vs {want}"
        assert bool(np_(done)[b]) == bool(g["done"][s]) and bool(np_(trunc)[b]) == bool(g["trunc"][s]), tag + " done/trunc"
        np.testing.assert_array_equal(met[b], g["metrics"][s], err_msg=tag + " metrics")
        if tagging:
            np.testing.assert_array_equal(np_(env.used_tag_actions)[b], g["used"][s].astype(bool), err_msg=tag + " used")
            np.testing.assert_array_equal(np_(env.tag_counts)[b], g["counts"][s], err_msg=tag + " counts")
            assert env.tag_reset_interval - int(np_(env._timer)[b]) == int(g["timer_left"][s]), tag + " timer"


# ------------------------------------------------------------------------------------------------
# (a) HIP kernels vs the reference's golden traces, fed numpy's MT19937 words as the tape
# ------------------------------------------------------------------------------------------------
KERNELS = ["compiled", "generic"]  # the compiled-in (Spec) kernels bench.py times, and the generic LDS-table kernels


def select_kernels(monkeypatch, kernels):
    """Route a handle's step launches: 'compiled' = whatever pick_spec selects (Spec<...> / SpecA<A> where one exists),
    'generic' = the generic kernels for every configuration (SUSNET_FORCE_GENERIC is read when the handle is created)."""
    if kernels == "generic":
        monkeypatch.setenv("SUSNET_FORCE_GENERIC", "1")
    else:
        monkeypatch.delenv("SUSNET_FORCE_GENERIC", raising=False)


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_hip_replays_reference_traces(pkg, family, kernels, monkeypatch):
    select_kernels(monkeypatch, kernels)
    gs = [load_golden(f"{GOLDEN_DIR}/{n}.npz") for n in FAMILIES[family]]
    meta = gs[0]["meta"]
    B = len(gs)
    tagging = meta["class"] == "tagging"
    env = env_from_meta(pkg, meta, B, rng="numpy", reward_dtype=torch.float64, tape_words=1 << 15)
    env._reseed([g["meta"]["seed"] for g in gs])  # env b consumes numpy's legacy stream of its fixture's seed
    env.reset()
    np.testing.assert_array_equal(np_(env.rng_cursor()), [g["words"][0] for g in gs])
    sampled = meta["mode"] == "sampled"
    S = len(gs[0]["done"])
    for s in range(S):
        if s > 0:
            mask = np.array([bool(g["ep_start"][s]) for g in gs])
            if mask.any():
                env.reset(mask=mask)
        if not sampled:
            for b, g in enumerate(gs):
                assert B == 1
                if g["inject"][s]:
                    kw = dict(agent_positions=g["pre_pos"][s][None], alive_agents=g["pre_alive"][s][None], t=g["pre_t"][s:s + 1])
                    if env.n_jobs:
                        kw.update(job_positions=g["pre_jobpos"][s][None], completed_jobs=g["pre_jobdone"][s][None])
                    if tagging:
                        kw.update(used_tag_actions=g["pre_used"][s][None], tag_counts=g["pre_counts"][s][None],
                                  tag_reset_timer=g["pre_timer"][s:s + 1])
                    env.set_state(**kw)
        # state before the step (after reset / injection) -- includes the reset's random spawns
        np.testing.assert_array_equal(np_(env.agent_positions), np.stack([g["pre_pos"][s] for g in gs]), err_msg=f"pre_pos {s}")
        np.testing.assert_array_equal(np_(env.imposter_mask), np.stack([g["pre_imp"][s] for g in gs]).astype(bool))
        if env.n_jobs:
            np.testing.assert_array_equal(np_(env.job_positions), np.stack([g["pre_jobpos"][s] for g in gs]))
        want_a = np.stack([g["actions"][s] for g in gs])
        if sampled:
            a = env.sample_actions().clone()
            np.testing.assert_array_equal(np_(a), want_a, err_msg=f"sampled actions step {s}")
        else:
            a = torch.as_tensor(want_a.astype(np.int64))
        _, rew, done, trunc, _ = env.step(a)
        check_step_against_golden(env, gs, s, rew, done, trunc, tagging)
    # raw words consumed == what numpy consumed (only comparable where no reset follows)
    cur = np_(env.rng_cursor())
    for b, g in enumerate(gs):
        if not (g["done"][-1] or g["trunc"][-1]):
            assert cur[b] == g["words"][-1], g["name"]


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("name", crc_names())
def test_hip_matches_reference_crc_streams(pkg, name, kernels, monkeypatch):
    """64 numpy seeds per configuration at once (one env per seed, TAPE mode fed numpy's own words), through the step API.
    The CRC of a step that ENDS an episode is taken from the fixture (with auto-reset the exported state is already the next
    episode's): terminal steps are compared field by field in test_hip_replays_reference_traces (no auto-reset) and, tick by
    tick against the oracle, in the fused test below."""
    import zlib

    select_kernels(monkeypatch, kernels)

    g = load_golden(f"{GOLDEN_DIR}/{name}.npz")
    meta = g["meta"]
    seeds, S = meta["seeds"], meta["n_steps"]
    B = len(seeds)
    tagging = meta["class"] == "tagging"
    env = env_from_meta(pkg, meta, B, rng="numpy", reward_dtype=torch.float64, tape_words=1 << 14, auto_reset=True)
    env._reseed(seeds)
    env.reset()
    got = np.zeros((B, S), dtype=np.uint32)
    for s in range(S):
        a = env.sample_actions().clone()
        _, rew, done, trunc, _ = env.step(a)
        # with auto-reset the exported state is already the next episode's; the record needs the terminal state,
        # which for an ended env is fully determined by the golden stream itself -> compare only non-ended envs'
        # state here and every env's actions / rewards / flags / info (the info counters stay readable)
        an, rn, dn, tn, mn = np_(a), np_(rew).astype(np.float64), np_(done), np_(trunc), np_(env._metrics)
        pos, alive, jd = np_(env.agent_positions), np_(env.alive_agents), np_(env.completed_jobs)
        used, counts, timer = np_(env.used_tag_actions), np_(env.tag_counts), np_(env._timer)
        for b in range(B):
            if dn[b] or tn[b]:
                got[b, s] = g["crc"][b, s]  # terminal state is gone after the in-launch reset: checked via the next steps
                continue
            rec = step_record_bytes(actions=an[b], pos=pos[b], alive=alive[b], jobdone=jd[b], rewards=rn[b], done=dn[b], trunc=tn[b],
                                    metrics=mn[b, :13], used=used[b] if tagging else None, counts=counts[b] if tagging else None,
                                    timer_left=env.tag_reset_interval - timer[b] if tagging else None)
            got[b, s] = zlib.crc32(rec)
    env.poll_errors()
    bad = np.argwhere(got != g["crc"])
    assert bad.size == 0, f"{name}: first mismatch at (seed index, step) = {bad[0].tolist()}"


def _tape_layouts():
    # (the compact 16-byte record exists for the 1v1 game: k_rollout_duel, on the empty grid and -- round 5 -- on the wall map)
    return [(n, lay) for n in crc_names() for lay in ("separate", "packed") + (("compact",) if n.startswith("crc_itg_1v1_") else ())]


@pytest.mark.parametrize("epw", [16, 32, 64])
@pytest.mark.parametrize("name,layout", _tape_layouts())
def test_fused_rollout_on_numpy_tapes_matches_reference_crc_streams(pkg, oracle_mod, name, layout, epw, monkeypatch):
    """The FUSED rollout kernels fed numpy's own MT19937 words, through the very instantiations bench.py times: trajectory as
    separate tensors (OUT_TRAJ_RAW8) and as PACKED RECORDS (OUT_RECORD: the headline layout), at 16, 32 and 64 environments
    per wave (SUSNET_EPW; 32 routes the 8-agent game to the two-lanes-per-environment kernel) -- k_rollout_duel<TapeRng, *>,
    k_rollout_swar<*, *, TapeRng>, k_rollout_swar2<*, *, TapeRng>, k_rollout<*, *, TapeRng>, whichever the configuration
    selects.  ONE launch of 96 ticks per fixture: sample_actions, step, in-launch reset all draw from the tape with numpy
    semantics, so env b reproduces the reference seeded with seed b.  Every tick's actions, rewards, done, truncated and
    post-reset raw observation must equal the oracle's MT19937 run (itself equal to the reference on these very seeds,
    tests/test_oracle_checksums.py), and for every step that did not end an episode the CRC of the step record rebuilt from
    the kernel's outputs (info counters taken from the oracle: the trajectory does not carry them) must equal the
    reference's.  (A step that ENDS an episode takes its CRC from the fixture: the in-launch reset has replaced the terminal
    state by then; its actions / rewards / flags are compared through the oracle above, and terminal states field by field in
    test_hip_replays_reference_traces.)"""
    import zlib

    g = load_golden(f"{GOLDEN_DIR}/{name}.npz")
    meta = g["meta"]
    seeds, S = meta["seeds"], meta["n_steps"]
    B = len(seeds)
    tagging = meta["class"] == "tagging"
    monkeypatch.setenv("SUSNET_EPW", str(epw))
    env = env_from_meta(pkg, meta, B, rng="numpy", tape_words=1 << 14, auto_reset=True, check_errors=False)
    lay = env.native_layout()
    assert lay.envs_per_wave == epw and lay.test_overrides == 2  # SUSNET_OVERRIDE_EPW, read when the handle was created
    packed = {"separate": False, "packed": True, "compact": "compact"}[layout]  # (compact: the 16-byte record of the 1v1 no-walls kernel)
    if packed and env.record_layout(packed) is None:
        pytest.skip("no packed record of this format for the configuration")
    if not packed and epw == 32 and env.record_layout() is None:
        pytest.skip("generic kernels: 16 and 64 environments per wave cover the lane masking")
    env._reseed(seeds)
    env.reset()
    traj = env.rollout(S, obs=pkg.ObsConfig("raw", dtype=torch.uint8), packed=packed)
    torch.cuda.synchronize()
    env.poll_errors()
    acts, rews, dones, truncs, obs = (np_(traj[k]) for k in ("actions", "rewards", "done", "truncated", "obs"))
    ob = oracle_mod.OracleBatch(oracle_mod.config_from_fixture_meta(meta), B)
    ob.seed_mt(seeds)
    ob.reset()
    A, J = env.n_agents, env.n_jobs
    interval = ob.envs[0].cfg.tag_reset_interval
    got = np.zeros((B, S), dtype=np.uint32)
    for s in range(S):
        oa = ob.sample_actions()
        np.testing.assert_array_equal(acts[s], oa, err_msg=f"{name} actions tick {s}")
        orew, odone, otrunc, rc = ob.step(oa)
        assert rc == 0
        assert np.array_equal(rews[s].astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"{name} rewards tick {s}"
        np.testing.assert_array_equal(dones[s], odone.astype(bool), err_msg=f"{name} done tick {s}")
        np.testing.assert_array_equal(truncs[s], otrunc.astype(bool), err_msg=f"{name} truncated tick {s}")
        met = ob.export()["metrics"]
        ended = (odone | otrunc).astype(bool)
        ob.reset(mask=ended)
        np.testing.assert_array_equal(obs[s], ob.obs_raw_u8(), err_msg=f"{name} raw obs tick {s}")
        for b in range(B):
            if ended[b]:
                got[b, s] = g["crc"][b, s]  # the terminal state is gone after the in-launch reset (checked through the oracle above)
                continue
            row = obs[s, b].astype(np.int64)
            k = 0
            pos = row[k:k + 2 * A].reshape(A, 2); k += 2 * A
            alive = row[k:k + A]; k += A
            jd = np.zeros(0, dtype=np.int64)
            if J > 0 or tagging:
                k += 2 * J
                jd = row[k:k + J]; k += J
            used = counts = left = None
            if tagging:
                used = row[k:k + A]; k += A
                counts = row[k:k + A]; k += A
                left = int(row[k])
            rec = step_record_bytes(actions=acts[s, b], pos=pos, alive=alive, jobdone=jd, rewards=rews[s, b].astype(np.float64),
                                    done=dones[s, b], trunc=truncs[s, b], metrics=met[b, :13], used=used, counts=counts, timer_left=left)
            got[b, s] = zlib.crc32(rec)
    bad = np.argwhere(got != g["crc"])
    assert bad.size == 0, f"{name}: first mismatch at (seed index, step) = {bad[0].tolist()}"
    # words consumed: the kernel's cursors equal the oracle's MT19937 word counts
    np.testing.assert_array_equal(np_(env.rng_cursor()).astype(np.uint64), ob.export()["cursor"], err_msg=f"{name} words consumed")


@pytest.mark.parametrize("family", [f for f in sorted(FAMILIES) if f.startswith("itg_1v1_")])
def test_production_1v1_kernel_steps_reference_states(pkg, family):
    """The kernel bench.py's headline number runs (k_step / k_rollout<PhiloxRng, Spec<2,0,ITG,...>>: branch-free duel) against
    the reference's own traces.  A 1v1 step with a fixed order draws nothing, so the production kernel can be fed the
    reference's pre-step states and actions directly: every recorded step of a trace becomes one env of a batch (state,
    t and info counters injected), ONE step launch, and the post-step state / rewards (bit patterns) / done / truncated /
    info must equal what the reference produced."""
    for n in FAMILIES[family]:
        g = load_golden(f"{GOLDEN_DIR}/{n}.npz")
        meta = g["meta"]
        S = len(g["done"])
        env = env_from_meta(pkg, meta, S, rng="philox", reward_dtype=torch.float64)
        env.reset()
        assert bool(np_(env.imposter_mask)[:, 0].all()) and not np_(env.imposter_mask)[:, 1].any()
        pre_metrics = np.zeros((S, 13), dtype=np.int64)
        for s in range(1, S):
            if not g["ep_start"][s]:
                pre_metrics[s] = g["metrics"][s - 1]
        env.set_state(agent_positions=g["pre_pos"], alive_agents=g["pre_alive"], t=g["pre_t"], metrics=pre_metrics)
        _, rew, done, trunc, _ = env.step(torch.as_tensor(g["actions"].astype(np.int64)))
        np.testing.assert_array_equal(np_(env.agent_positions), g["pos"], err_msg=n + " pos")
        np.testing.assert_array_equal(np_(env.alive_agents), g["alive"].astype(bool), err_msg=n + " alive")
        got = np_(rew).astype(np.float64)
        assert got.view(np.uint64).tolist() == g["rewards"].view(np.uint64).tolist(), n + " rewards"
        np.testing.assert_array_equal(np_(done), g["done"].astype(bool), err_msg=n + " done")
        np.testing.assert_array_equal(np_(trunc), g["trunc"].astype(bool), err_msg=n + " trunc")
        np.testing.assert_array_equal(np_(env._metrics), g["metrics"], err_msg=n + " info")


# ------------------------------------------------------------------------------------------------
# (b) HIP kernels vs the CPU oracle on the production (Philox) stream, with auto-reset
# ------------------------------------------------------------------------------------------------
CONFIGS = {
    "itg_1v1_nowalls": dict(cls="itg", kw=dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                                              time_step_reward=0, include_walls=False), n=9),
    "itg_1v1_walls": dict(cls="itg", kw=dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                                            time_step_reward=0), n=9),
    # imposter slot drawn per episode: not the compiled-in 1v1 kernel (roles are data there)
    "itg_1v1_shuffle": dict(cls="itg", kw=dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=2,
                                              time_step_reward=0, shuffle_imposter_index=True), n=9),
    "base_1v2_j4_14": dict(cls="base", kw=dict(n_imposters=1, n_crew=2, n_jobs=4), n=14),
    "base_2v6_j4_14": dict(cls="base", kw=dict(n_imposters=2, n_crew=6, n_jobs=4), n=14),
    "base_3v9_j8_16": dict(cls="base", kw=dict(n_imposters=3, n_crew=9, n_jobs=8, max_time_steps=60), n=16),
    # the ABI's maximum sizes: 16 agents, 16 jobs, 16x16 grid (generic LDS-table kernels)
    "base_7v9_j16_16": dict(cls="base", kw=dict(n_imposters=7, n_crew=9, n_jobs=16, max_time_steps=40), n=16),
    "tagging_1v4_j5": dict(cls="tagging", kw=dict(n_imposters=1, n_crew=4, n_jobs=5, tag_reset_interval=6), n=9),
    "tagging_2v6_j4_14": dict(cls="tagging", kw=dict(n_imposters=2, n_crew=6, n_jobs=4, tag_reset_interval=11, time_step_reward=-1), n=14),
    "itg_1v5_j3": dict(cls="itg", kw=dict(n_crew=5, n_jobs=3, kill_reward=-3, sabotage_reward=1, end_of_game_reward=7,
                                         time_step_reward=-1, shuffle_imposter_index=True), n=9),
    # notebooks/visualizing_games.ipynb: ImposterTrainingGround(n_crew=10): 11 agents = three words of agent bytes (round 5)
    "itg_1v10": dict(cls="itg", kw=dict(n_crew=10, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0, time_step_reward=0), n=9),
}


def make_pair(pkg, oracle_mod, name, B, seed, env_id_base=0, **envkw):
    spec = CONFIGS[name]
    kw = dict(spec["kw"])
    walls = kw.pop("include_walls", True)
    grid = pkg.four_room_grid(spec["n"], walls)
    cls = {"itg": pkg.BatchedImposterTrainingGround, "base": pkg.BatchedFourRoomEnv, "tagging": pkg.BatchedFourRoomEnvWithTagging}[spec["cls"]]
    env = cls(**kw, grid=grid, batch=B, rng="philox", seed=seed, env_id_base=env_id_base, **envkw)
    okw = dict(kw)
    if spec["cls"] == "itg":
        okw.setdefault("shuffle_imposter_index", False)
    ob = oracle_mod.OracleBatch(oracle_mod.make_config(spec["cls"], grid=grid.astype(np.uint8), **okw), B)
    ob.set_philox(seed, env_id_base, 0)
    return env, ob


def compare_full_state(env, ob, tag):
    e = ob.export()
    np.testing.assert_array_equal(np_(env.agent_positions), e["pos"], err_msg=tag + " pos")
    np.testing.assert_array_equal(np_(env.alive_agents), e["alive"].astype(bool), err_msg=tag + " alive")
    np.testing.assert_array_equal(np_(env.imposter_mask), e["imp"].astype(bool), err_msg=tag + " imp")
    if env.n_jobs:
        np.testing.assert_array_equal(np_(env.job_positions), e["jobpos"], err_msg=tag + " jobpos")
        np.testing.assert_array_equal(np_(env.completed_jobs), e["jobdone"].astype(bool), err_msg=tag + " jobdone")
    np.testing.assert_array_equal(np_(env.t), e["t"], err_msg=tag + " t")
    np.testing.assert_array_equal(np_(env.rng_cursor()).astype(np.uint64), e["cursor"], err_msg=tag + " cursor")
    np.testing.assert_array_equal(np_(env.episode_index()).astype(np.uint32), e["episode"], err_msg=tag + " episode index (RESET stream)")
    if env.VARIANT == 2:
        np.testing.assert_array_equal(np_(env.used_tag_actions), e["used"].astype(bool), err_msg=tag + " used")
        np.testing.assert_array_equal(np_(env.tag_counts), e["counts"], err_msg=tag + " counts")
        np.testing.assert_array_equal(np_(env._timer), e["timer"], err_msg=tag + " timer")


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_hip_matches_oracle_philox_autoreset(pkg, oracle_mod, name):
    B, steps, seed = 2048 + 37, 150, 77
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, env_id_base=10_000_000_000, auto_reset=True, check_errors=False)
    env.reset()
    ob.reset()
    compare_full_state(env, ob, f"{name} reset")
    episodes = 0
    for s in range(steps):
        a = env.sample_actions().clone()
        oa = ob.sample_actions()
        np.testing.assert_array_equal(np_(a), oa, err_msg=f"{name} sample_actions step {s}")
        _, rew, done, trunc, _ = env.step(a)
        orew, odone, otrunc, rc = ob.step(oa)
        assert rc == 0
        got = np_(rew).astype(np.float64)
        assert np.array_equal(got.view(np.uint64), orew.view(np.uint64)), f"{name} rewards step {s}"
        np.testing.assert_array_equal(np_(done), odone.astype(bool), err_msg=f"{name} done step {s}")
        np.testing.assert_array_equal(np_(trunc), otrunc.astype(bool), err_msg=f"{name} trunc step {s}")
        # info of the terminal step stays readable until the next step (lazy metric reset)
        np.testing.assert_array_equal(np_(env._metrics), ob.export()["metrics"], err_msg=f"{name} metrics step {s}")
        ended = (odone | otrunc).astype(bool)
        episodes += int(ended.sum())
        ob.reset(mask=ended)
        if s % 10 == 0 or s == steps - 1:
            compare_full_state(env, ob, f"{name} step {s}")
    env.poll_errors()
    life = np_(env.lifetime_totals())
    assert life[0] == episodes, "lifetime episode count"


def test_imported_dead_agents_in_the_compiled_1v1_kernel(pkg, oracle_mod):
    """The 1v1 kernel takes its rewards from a few table constants selected by {kill landed, crew dead, agent dead}; states a
    caller assigns (dead imposter, dead crew, both) must behave like the reference's general reward chain (oracle)."""
    name, B = "itg_1v1_nowalls", 256
    env, ob = make_pair(pkg, oracle_mod, name, B, 19, auto_reset=False, check_errors=True)
    env.reset()
    ob.reset()
    alive = np.ones((B, 2), dtype=bool)
    alive[0::4, 0] = False            # dead imposter
    alive[1::4, 1] = False            # dead crew member (a finished episode)
    alive[2::4] = False               # both
    env.set_state(alive_agents=alive)
    for b in range(B):
        ob.set_state(b, alive=alive[b].astype(np.uint8))
    for s in range(25):
        a = env.sample_actions().clone()
        oa = ob.sample_actions()
        np.testing.assert_array_equal(np_(a), oa)
        _, rew, done, trunc, _ = env.step(a)
        orew, odone, otrunc, _ = ob.step(oa)
        assert np.array_equal(np_(rew).astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"rewards step {s}"
        np.testing.assert_array_equal(np_(done), odone.astype(bool))
        np.testing.assert_array_equal(np_(trunc), otrunc.astype(bool))
    compare_full_state(env, ob, "after stepping imported states")


@pytest.mark.parametrize("name", ["itg_1v1_nowalls", "base_1v2_j4_14", "base_2v6_j4_14"])
def test_hip_matches_oracle_without_reset_after_done(pkg, oracle_mod, name):
    """Callers may keep stepping a finished episode (the reference allows it): compiled-in kernels included."""
    B, steps, seed = 512, 220, 13
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=False, check_errors=False)
    env.reset()
    ob.reset()
    for s in range(steps):
        a = env.sample_actions().clone()
        oa = ob.sample_actions()
        np.testing.assert_array_equal(np_(a), oa)
        _, rew, done, trunc, _ = env.step(a)
        orew, odone, otrunc, rc = ob.step(oa)
        assert np.array_equal(np_(rew).astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"{name} rewards step {s}"
        np.testing.assert_array_equal(np_(done), odone.astype(bool))
        np.testing.assert_array_equal(np_(env._metrics), ob.export()["metrics"])
    compare_full_state(env, ob, f"{name} coasting")


@pytest.mark.parametrize("B", [1, 63, 64, 65])
def test_tiny_and_ragged_batches(pkg, oracle_mod, B):
    """One env, one lane short of a wave, exactly one wave, one lane over: step API and fused rollout."""
    for name in ("itg_1v1_nowalls", "base_2v6_j4_14", "base_7v9_j16_16"):
        env, ob = make_pair(pkg, oracle_mod, name, B, 3, auto_reset=True, check_errors=True)
        env.reset()
        ob.reset()
        for s in range(12):
            a = env.sample_actions().clone()
            oa = ob.sample_actions()
            np.testing.assert_array_equal(np_(a), oa)
            _, rew, done, trunc, _ = env.step(a)
            orew, odone, otrunc, _ = ob.step(oa)
            assert np.array_equal(np_(rew).astype(np.float64).view(np.uint64), orew.view(np.uint64))
            ob.reset(mask=(odone | otrunc).astype(bool))
        traj = env.rollout(9, obs=pkg.ObsConfig("raw", dtype=torch.uint8))
        for s in range(9):
            oa = ob.sample_actions()
            np.testing.assert_array_equal(np_(traj["actions"])[s], oa)
            orew, odone, otrunc, _ = ob.step(oa)
            assert np.array_equal(np_(traj["rewards"])[s].astype(np.float64).view(np.uint64), orew.view(np.uint64))
            ob.reset(mask=(odone | otrunc).astype(bool))
            np.testing.assert_array_equal(np_(traj["obs"])[s], ob.obs_raw_u8())
        env._export(full=True)
        compare_full_state(env, ob, f"{name} B={B}")


def test_graph_replayed_random_steps_match_oracle(pkg, oracle_mod):
    """hipGraph of {sample_actions; step}: the step counter lives in device memory, so every replay draws the next
    tick's actions.  Each replay is checked against the oracle; a fused rollout afterwards continues the same streams
    from the device counter, and turning device mode off hands the count back to the host."""
    for name in ("itg_1v1_nowalls", "base_1v2_j4_14"):
        B = 3000
        env, ob = make_pair(pkg, oracle_mod, name, B, 23, auto_reset=True, check_errors=False, export_state=False)
        env.reset()
        ob.reset()
        graph4 = env.capture_random_step(4)     # warm-up: 2 ticks (capturing itself executes nothing)
        graph4.replay()                         # 4 ticks in one replay
        torch.cuda.synchronize()
        for _ in range(2 + 4):                  # (the oracle follows: 2 warm-up ticks + the replayed 4)
            oa = ob.sample_actions()
            _, odone, otrunc, _ = ob.step(oa)
            ob.reset(mask=(odone | otrunc).astype(bool))
        graph = env.capture_random_step()       # 2 more warm-up ticks, then a 1-tick graph
        for _ in range(2):
            oa = ob.sample_actions()
            _, odone, otrunc, _ = ob.step(oa)
            ob.reset(mask=(odone | otrunc).astype(bool))
        acts_buf = env._actions_view
        for s in range(40):
            graph.replay()
            torch.cuda.synchronize()
            oa = ob.sample_actions()
            np.testing.assert_array_equal(np_(acts_buf), oa, err_msg=f"{name} replay {s}")
            orew, odone, otrunc, _ = ob.step(oa)
            assert np.array_equal(np_(env._rewards_view).astype(np.float64).view(np.uint64), orew.view(np.uint64))
            np.testing.assert_array_equal(np_(env._done), odone.astype(bool))
            ob.reset(mask=(odone | otrunc).astype(bool))
        assert env.tick == 48
        traj = env.rollout(9, obs=pkg.ObsConfig("raw", dtype=torch.uint8))   # device counter feeds the fused kernel too
        torch.cuda.synchronize()
        for s in range(9):
            oa = ob.sample_actions()
            np.testing.assert_array_equal(np_(traj["actions"])[s], oa, err_msg=f"{name} rollout tick {s} after replays")
            orew, odone, otrunc, _ = ob.step(oa)
            ob.reset(mask=(odone | otrunc).astype(bool))
            np.testing.assert_array_equal(np_(traj["obs"])[s], ob.obs_raw_u8())
        assert env.tick == 57
        env.device_tick(False)
        assert env.tick == 57
        a = env.sample_actions().clone()
        np.testing.assert_array_equal(np_(a), ob.sample_actions())
        env._export(full=True)
        compare_full_state(env, ob, f"{name} after graph replays")


def test_two_rank_bench_line_equals_one_process_over_the_same_env_ids(pkg):
    """bench.py's N > 1 path end to end on this one-GPU box: `bench.launch_command(2, ...)` started as a CHILD process (two ranks under
    torch.distributed.run, gloo rendezvous on 127.0.0.1, both ranks on device 0: SUSNET_BENCH_BACKEND / SUSNET_BENCH_ONE_DEVICE) -- rank 1
    executes the timed path, the all-gather of the episode metrics and the per-rank launch times.  The JSON line's episode metrics must
    equal a one-process run over the same GLOBAL env ids (batch 2 x 4096, same seed, same launches): sharding is invisible."""
    import importlib.util
    import json
    import os
    import socket
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    Bshard, K, W, T = 4096, 3, 1, 64
    flags = ["--steps", str(K), "--warmup", str(W), "--ticks", str(T), "--config", "cfg3", "--no-cpu-baseline", "--no-secondary", "--repeats", "0", "--settle-ms", "0"]
    env = dict(os.environ, SUSNET_BENCH_BACKEND="gloo", SUSNET_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(bench.launch_command(2, port, ["--gpus", "2", "--batch", str(Bshard)] + flags), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, "rank 0 prints ONE line"
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 2 * Bshard and two["scaling"] == "weak"
    per_rank = two["roofline"]["avg_launch_us_per_rank"]
    assert len(per_rank["all"]) == 2 and per_rank["min"] > 0
    assert two["episode_metrics"]["env_steps"] == 2 * Bshard * T * (K + W + 8)  # (warm-up + timed + the 8 event-pair launches)
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", str(2 * Bshard)] + flags, env=dict(os.environ),
                       capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-2000:]
    one = json.loads([ln for ln in q.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert one["episode_metrics"] == two["episode_metrics"], (one["episode_metrics"], two["episode_metrics"])
    assert one["episode_metrics"]["episodes"] > 0


def test_node_metrics_over_rccl(pkg):
    """The one collective of the path (dist.node_metrics: device reduction + all_gather_into_tensor) through the real
    RCCL backend, single rank (a one-GPU box cannot host two RCCL ranks; the two-rank control flow is the gloo test)."""
    import torch.distributed as dist

    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
        created = True
    try:
        env = pkg.BatchedImposterTrainingGround(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                                                time_step_reward=0, include_walls=False, batch=4096, auto_reset=True, seed=2,
                                                max_time_steps=50)
        env.reset()
        traj = env.rollout(200)
        torch.cuda.synchronize()
        m = pkg.dist.node_metrics(env)
        ended = int((traj["done"] | traj["truncated"]).sum())
        assert m["episodes"] == ended and m["per_rank_episodes"] == [ended]
        assert m["imposter_won"] == int(traj["done"].sum()) and m["truncated"] == int((traj["truncated"] & ~traj["done"]).sum()) \
            or m["truncated"] == int(traj["truncated"].sum())
    finally:
        if created:
            dist.destroy_process_group()


def test_sharding_is_invisible(pkg, oracle_mod):
    """Env b of a shard with env_id_base = k behaves exactly as env k + b of one big batch."""
    name, B, seed = "base_2v6_j4_14", 1024, 5
    big, _ = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False)
    lo, _ = make_pair(pkg, oracle_mod, name, B // 2, seed, env_id_base=0, auto_reset=True, check_errors=False)
    hi, _ = make_pair(pkg, oracle_mod, name, B // 2, seed, env_id_base=B // 2, auto_reset=True, check_errors=False)
    for e in (big, lo, hi):
        e.reset()
    for s in range(60):
        outs = []
        for e in (big, lo, hi):
            a = e.sample_actions().clone()
            _, rew, done, trunc, _ = e.step(a)
            outs.append((np_(a), np_(rew), np_(done), np_(trunc), np_(e.agent_positions)))
        for k in range(5):
            np.testing.assert_array_equal(outs[0][k], np.concatenate([outs[1][k], outs[2][k]]), err_msg=f"step {s} field {k}")
    tot = np_(big.lifetime_totals())
    np.testing.assert_array_equal(tot, np_(lo.lifetime_totals()) + np_(hi.lifetime_totals()))


@pytest.mark.parametrize("name", ["itg_1v1_nowalls", "itg_1v1_walls", "itg_1v1_shuffle", "base_1v2_j4_14", "base_2v6_j4_14", "tagging_1v4_j5", "itg_1v5_j3",
                                  "itg_1v10", "base_3v9_j8_16"])
def test_fused_rollout_matches_oracle(pkg, oracle_mod, name):
    B, T, seed = 1000, 120, 9
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False)
    env.reset()
    ob.reset()
    traj = env.rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8))
    torch.cuda.synchronize()
    acts, rews, dones, truncs, obs = (np_(traj[k]) for k in ("actions", "rewards", "done", "truncated", "obs"))
    for s in range(T):
        oa = ob.sample_actions()
        np.testing.assert_array_equal(acts[s], oa, err_msg=f"{name} actions tick {s}")
        orew, odone, otrunc, rc = ob.step(oa)
        assert np.array_equal(rews[s].astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"{name} rewards tick {s}"
        np.testing.assert_array_equal(dones[s], odone.astype(bool))
        np.testing.assert_array_equal(truncs[s], otrunc.astype(bool))
        if s == T - 1:
            last_info = ob.export()["metrics"].copy()  # info of the launch's last tick, before any reset
        ob.reset(mask=(odone | otrunc).astype(bool))
        np.testing.assert_array_equal(obs[s], ob.obs_raw_u8(), err_msg=f"{name} raw obs tick {s}")
    env._export(full=True)
    compare_full_state(env, ob, f"{name} after rollout")
    # the info counters of the last tick stay readable (also for envs whose episode ended on it)
    np.testing.assert_array_equal(np_(env._metrics), last_info, err_msg=f"{name} info after rollout")
    # further launches continue the same streams: odd lengths and odd starting ticks (the 1v1 kernels walk the
    # action stream in (even, odd) tick pairs), with and without the trajectory-mode outputs
    for n, with_obs in ((7, False), (5, True), (1, True), (2, True), (3, False), (4, True)):
        traj2 = env.rollout(n, obs=pkg.ObsConfig("raw", dtype=torch.uint8) if with_obs else None)
        torch.cuda.synchronize()
        for s in range(n):
            oa = ob.sample_actions()
            np.testing.assert_array_equal(np_(traj2["actions"])[s], oa, err_msg=f"{name} launch of {n}, tick {s}")
            orew, odone, otrunc, _ = ob.step(oa)
            assert np.array_equal(np_(traj2["rewards"])[s].astype(np.float64).view(np.uint64), orew.view(np.uint64))
            np.testing.assert_array_equal(np_(traj2["done"])[s], odone.astype(bool))
            ob.reset(mask=(odone | otrunc).astype(bool))
            if with_obs:
                np.testing.assert_array_equal(np_(traj2["obs"])[s], ob.obs_raw_u8())
    env._export(full=True)
    compare_full_state(env, ob, f"{name} after the short launches")


@pytest.mark.parametrize("name", ["itg_1v1_nowalls", "itg_1v1_walls", "base_1v2_j4_14", "base_2v6_j4_14", "tagging_1v4_j5", "itg_1v10", "base_3v9_j8_16"])
def test_packed_record_rollout_matches_oracle(pkg, oracle_mod, name, monkeypatch):
    """The packed per-env-step record (one wide store per lane) holds the same fields as the separate trajectory tensors;
    its strided views are checked against the oracle loop, across launches of odd lengths and a chunked launch."""
    B, seed = 1000 if name != "base_2v6_j4_14" else 40000, 9
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False)
    if env.record_layout() is None:
        pytest.skip("configuration not compiled in")
    lay = env.record_layout()
    segs = [tuple(lay.obs_segments[k]) for k in range(lay.n_obs_segments)]  # (family records: the observation in up to four segments)
    assert lay.record_bytes % 4 == 0 and sum(n for _, n in segs) == env.flattened_state_size and lay.off_truncated + 1 <= lay.record_bytes
    env.reset()
    ob.reset(threads=0)
    obs_cfg = pkg.ObsConfig("raw", dtype=torch.uint8)
    for n in (24, 5, 1, 2, 3):
        env.set_launch_limit(2 * B * lay.record_bytes + 8 if n == 5 else 0)  # n == 5: 2 ticks per launch
        traj = env.rollout(n, obs=obs_cfg, packed=True)
        torch.cuda.synchronize()
        assert traj["rewards"].shape == (n, B, env.n_agents) and traj["obs"].shape == (n, B, env.flattened_state_size)
        for s in range(n):
            oa = ob.sample_actions()
            np.testing.assert_array_equal(np_(traj["actions"])[s], oa, err_msg=f"{name} launch of {n}, tick {s}")
            orew, odone, otrunc, _ = ob.step(oa, threads=0)
            assert np.array_equal(np_(traj["rewards"])[s].astype(np.float64).view(np.uint64), orew.view(np.uint64))
            np.testing.assert_array_equal(np_(traj["done"])[s], odone.astype(bool))
            np.testing.assert_array_equal(np_(traj["truncated"])[s], otrunc.astype(bool))
            ob.reset(mask=(odone | otrunc).astype(bool))
            np.testing.assert_array_equal(np_(traj["obs"])[s], ob.obs_raw_u8())
        used = np.zeros(lay.record_bytes, dtype=bool)  # (the field order is the layout's business: read it through the offsets)
        for off, n in [(lay.off_rewards, 4 * env.n_agents), (lay.off_actions, env.n_agents), (lay.off_done, 1), (lay.off_truncated, 1)] + segs:
            assert not used[off:off + n].any(), "record fields overlap"
            used[off:off + n] = True
        if lay.n_obs_segments == 1:  # (the family's record keeps eight job slots whatever the job count: unused slots hold zeros / stale cells)
            assert not np_(env.unpack_record(traj["record"]))[:, :, ~used].any(), "padding bytes are zero"
    env._export(full=True)
    compare_full_state(env, ob, f"{name} after packed rollouts")


def test_long_trajectory_launches_are_chunked(pkg, oracle_mod, monkeypatch):
    """Trajectory-mode kernels address outputs with 32-bit offsets; requests whose arrays would pass 2 GiB run as
    consecutive launches.  Forced here with a small limit: 7 ticks per launch, i.e. odd starts and a ragged tail."""
    name, B, T, seed = "itg_1v1_nowalls", 1000, 45, 5
    monkeypatch.setenv("SUSNET_TRAJ_MAX_BYTES", str(7 * 8 * B + 100))
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False)
    env.reset()
    ob.reset()
    traj = env.rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8))
    torch.cuda.synchronize()
    for s in range(T):
        oa = ob.sample_actions()
        np.testing.assert_array_equal(np_(traj["actions"])[s], oa, err_msg=f"actions tick {s}")
        orew, odone, otrunc, _ = ob.step(oa)
        assert np.array_equal(np_(traj["rewards"])[s].astype(np.float64).view(np.uint64), orew.view(np.uint64))
        np.testing.assert_array_equal(np_(traj["done"])[s], odone.astype(bool))
        np.testing.assert_array_equal(np_(traj["truncated"])[s], otrunc.astype(bool))
        ob.reset(mask=(odone | otrunc).astype(bool))
        np.testing.assert_array_equal(np_(traj["obs"])[s], ob.obs_raw_u8(), err_msg=f"raw obs tick {s}")
    env._export(full=True)
    compare_full_state(env, ob, "after chunked rollout")
    assert int(env.tick) == T


@pytest.mark.parametrize("name,B,packed", [("base_2v6_j4_14", 32768 + 96, False), ("itg_1v1_nowalls", 65536 + 32, False), ("base_1v2_j4_14", 65536 + 32, False),
                                           ("base_1v2_j4_14", 65536 + 32, True), ("base_2v6_j4_14", 32768, True), ("itg_1v1_nowalls", 65536, "compact"),
                                           ("itg_1v1_walls", 65536 + 32, "compact"), ("itg_1v1_walls", 65536 + 32, False), ("itg_1v1_walls", 2048, True),
                                           ("tagging_1v4_j5", 4096, False), ("itg_1v5_j3", 2048, False)])
def test_fused_rollout_large_batches_all_wave_widths(pkg, oracle_mod, name, B, packed):
    """Batches >= 32768 / 65536 run 32 / 64 environments per wave (smaller ones 16): same results.  The BASELINE configurations at the
    benchmark's batch (+ a ragged last wave), separate tensors and the records bench.py times, against the oracle (OpenMP batch)."""
    T, seed = 24, 31
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False)
    env.reset()
    ob.reset(threads=0)
    traj = env.rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8), packed=packed)
    torch.cuda.synchronize()
    acts, rews, dones, truncs, obs = (np_(traj[k]) for k in ("actions", "rewards", "done", "truncated", "obs"))
    for s in range(T):
        oa = ob.sample_actions()
        np.testing.assert_array_equal(acts[s], oa, err_msg=f"{name} actions tick {s}")
        orew, odone, otrunc, rc = ob.step(oa, threads=0)
        assert np.array_equal(rews[s].astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"{name} rewards tick {s}"
        np.testing.assert_array_equal(dones[s], odone.astype(bool))
        np.testing.assert_array_equal(truncs[s], otrunc.astype(bool))
        ob.reset(mask=(odone | otrunc).astype(bool))
        np.testing.assert_array_equal(obs[s], ob.obs_raw_u8(), err_msg=f"{name} raw obs tick {s}")
    env._export(full=True)
    compare_full_state(env, ob, f"{name} after rollout")


def test_two_lanes_per_env_rollout_equals_the_one_lane_kernel(pkg, oracle_mod, monkeypatch):
    """8-agent configurations at 32 environments per wave run with TWO lanes per environment (susnet_swar2.h).  Forced at a
    small batch here (SUSNET_EPW) and compared, output by output, with the one-lane kernel (itself pinned to the oracle
    above) over long launches with many episode ends: separate tensors + replay feed, packed records, no outputs; the first
    120 ticks also against the oracle."""
    name, B, seed = "base_2v6_j4_14", 1000, 77
    obs_cfg = pkg.ObsConfig("raw", dtype=torch.uint8)
    envs = {}
    for epw in (16, 32):
        monkeypatch.setenv("SUSNET_EPW", str(epw))
        envs[epw], ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False)
        envs[epw].reset()
    ob.reset()
    outs = {}
    for epw, env in envs.items():
        got = {}
        bufs = env.alloc_rollout(400, obs=obs_cfg, replay_feed=True)
        bufs["term_obs"].zero_()  # (written only where an episode ended)
        env.rollout_into(400, bufs)
        got["feed"] = {k: np_(bufs[k]).copy() for k in ("actions", "rewards", "done", "truncated", "obs", "term_obs", "roles")}
        env.rollout(37, store=(), obs=None)  # state-only fast-forward, odd length
        tr = env.rollout(203, obs=obs_cfg, packed=True)
        got["packed"] = {k: np_(tr[k]).copy() for k in ("actions", "rewards", "done", "truncated", "obs")}  # (views: the two kernels lay the record out differently)
        tr = env.rollout(50, obs=None)
        got["traj"] = {k: np_(tr[k]).copy() for k in ("actions", "rewards", "done", "truncated")}
        torch.cuda.synchronize()
        env._export(full=True)
        got["state"] = {k: np_(getattr(env, k)).copy() for k in ("agent_positions", "alive_agents", "imposter_mask", "job_positions", "completed_jobs", "t")}
        got["metrics"], got["cursor"] = np_(env._metrics).copy(), np_(env.rng_cursor()).copy()
        got["lifetime"] = np_(env.lifetime_totals()).copy()
        outs[epw] = got
    a, b = outs[16], outs[32]
    assert a["feed"]["done"].sum() > 50, "the launches must cross many episode ends"
    for grp in ("feed", "packed", "traj", "state"):
        for k in a[grp]:
            np.testing.assert_array_equal(a[grp][k], b[grp][k], err_msg=f"{grp}/{k}")
    for k in ("metrics", "cursor", "lifetime"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    f = b["feed"]
    for s in range(120):
        oa = ob.sample_actions()
        np.testing.assert_array_equal(f["actions"][s], oa, err_msg=f"actions tick {s}")
        orew, odone, otrunc, _ = ob.step(oa)
        assert np.array_equal(f["rewards"][s].astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"rewards tick {s}"
        np.testing.assert_array_equal(f["done"][s], odone.astype(bool))
        ended = (odone | otrunc).astype(bool)
        np.testing.assert_array_equal(f["term_obs"][s][ended], ob.obs_raw_u8()[ended], err_msg=f"terminal rows tick {s}")
        ob.reset(mask=ended)
        np.testing.assert_array_equal(f["obs"][s], ob.obs_raw_u8(), err_msg=f"raw obs tick {s}")


@pytest.mark.parametrize("name,obs", [("itg_1v5_j3", "flat"), ("base_2v6_j4_14", "flat"), ("base_2v6_j4_14", "planes"), ("tagging_2v6_j4_14", "planes"),
                                      ("tagging_1v4_j5", "planes"), ("itg_1v10", "flat"), ("base_3v9_j8_16", "planes"), ("itg_1v1_walls", "planes")])
def test_fused_rollout_with_float_observations_matches_oracle(pkg, oracle_mod, name, obs):
    """The OUT_ANY instantiations of the byte-parallel rollouts (any observation mode fused in: the largest kernels of the library -- round 5:
    back under 256 registers, no accumulator registers, the episode of a finishing lane drawn on the spot): a ragged last wave, many
    in-launch resets, the full trajectory and the float observation of every tick against the oracle.  The 11 / 12-agent games hand this
    mode to the generic kernel on the same state blob and streams; the 1v1 wall-map game runs its table kernel here."""
    B, T, seed = 64 * 9 + 23, 90, 13
    short = name + "@20"  # (episodes of at most 20 steps: every environment resets several times inside the launch)
    CONFIGS[short] = dict(CONFIGS[name], kw=dict(CONFIGS[name]["kw"], max_time_steps=20))
    try:
        env, ob = make_pair(pkg, oracle_mod, short, B, seed, auto_reset=True, check_errors=False)
    finally:
        del CONFIGS[short]
    env.reset()
    ob.reset()
    comps = ["onehot_pos", "alive_crew", "coord_pos"]
    cfg = pkg.ObsConfig("flat", comps) if obs == "flat" else pkg.ObsConfig("planes")
    ended = 0
    for n in (T, 7):  # (a second, short launch: starts mid-stream)
        traj = env.rollout(n, obs=cfg)
        torch.cuda.synchronize()
        got = traj["obs"] if not isinstance(traj["obs"], (tuple, list)) else traj["obs"][0]
        got2 = traj["obs"][1] if isinstance(traj["obs"], (tuple, list)) else traj.get("obs_non_spatial")
        for s in range(n):
            oa = ob.sample_actions()
            np.testing.assert_array_equal(np_(traj["actions"])[s], oa, err_msg=f"{name} actions tick {s}")
            orew, odone, otrunc, _ = ob.step(oa)
            assert np.array_equal(np_(traj["rewards"])[s].astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"{name} rewards tick {s}"
            np.testing.assert_array_equal(np_(traj["done"])[s], odone.astype(bool))
            np.testing.assert_array_equal(np_(traj["truncated"])[s], otrunc.astype(bool))
            e = (odone | otrunc).astype(bool)
            ended += int(e.sum())
            ob.reset(mask=e)
            if obs == "flat":
                assert np_(got[s]).view(np.uint32).tolist() == ob.obs_flat(comps).view(np.uint32).tolist(), f"{name} flat obs tick {s}"
            else:
                wsp, wnon = ob.obs_planes()
                np.testing.assert_array_equal(np_(got[s]), wsp, err_msg=f"{name} planes tick {s}")
                if got2 is not None:
                    np.testing.assert_array_equal(np_(got2[s]), wnon, err_msg=f"{name} non-spatial tick {s}")
    assert ended > B // 4, "the launches must cross episode ends"
    env._export(full=True)
    compare_full_state(env, ob, f"{name} after the float-observation rollouts")


@pytest.mark.parametrize("cls", ["base", "itg", "tagging"])
@pytest.mark.parametrize("A", [3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_family_sweep_matches_oracle(pkg, oracle_mod, cls, A):
    """The byte-parallel FAMILY instantiations (susnet_family.h: 3..12 agents -- tagging 3..8 -- x {FourRoomEnv, ImposterTrainingGround, tagging} x
    action order x 1..3 imposters, job count and role shuffle at run time; 9..12 agents: three words of agent bytes, round 5), each through BOTH trajectory layouts -- packed records and separate tensors with
    the raw observation -- against the oracle: job counts 0 / 3 / 8, fixed and shuffled roles, a ragged last wave, 12-step episodes."""
    if cls == "tagging" and A > 8:
        pytest.skip("tagging games above 8 agents run the generic kernels (the family's tag section pairs two words of agents)")
    B, T, n_cases = 64 * 2 + 3, 36, 0
    raw8 = pkg.ObsConfig("raw", dtype=torch.uint8)
    for n_imp in ((1,) if cls == "itg" else (1, 2, 3)):  # (round 5: three imposters -- the kill turns of three killers in turn order)
        if A - n_imp <= (0 if cls == "itg" else n_imp):  # (base.py:243-249: more crew members than imposters)
            continue
        for J in (0, 3, 8):
            for order_random in ((False,) if cls == "itg" else (True, False)):
                for shuffle in (False, True):
                    kw = dict(n_crew=A - n_imp, n_jobs=J, max_time_steps=12, shuffle_imposter_index=shuffle)
                    if cls == "itg":
                        kw.update(kill_reward=-3, sabotage_reward=1, end_of_game_reward=5, time_step_reward=-1)
                    else:
                        kw.update(n_imposters=n_imp, is_action_order_random=order_random)
                    if cls == "tagging":
                        kw.update(tag_reset_interval=5)
                    name = f"sweep {cls} {n_imp}v{A - n_imp} j{J} order{int(order_random)} shuffle{int(shuffle)}"
                    CONFIGS[name] = dict(cls=cls, kw=kw, n=9)
                    try:
                        for packed in (True, False):
                            env, ob = make_pair(pkg, oracle_mod, name, B, 40 + J, auto_reset=True, check_errors=False)
                            if packed and env.record_layout() is None:
                                continue
                            assert env.record_layout() is not None, f"{name}: every 3..12-agent game with at most 3 imposters and 8 jobs is in the family"
                            env.reset()
                            ob.reset()
                            traj = env.rollout(T, obs=raw8, packed=packed)
                            torch.cuda.synchronize()
                            acts, rews, dones, truncs, obs = (np_(traj[k]) for k in ("actions", "rewards", "done", "truncated", "obs"))
                            for s_ in range(T):
                                oa = ob.sample_actions()
                                assert np.array_equal(acts[s_], oa), f"{name} packed={packed} actions tick {s_}"
                                orew, odone, otrunc, _ = ob.step(oa)
                                assert np.array_equal(rews[s_].astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"{name} packed={packed} rewards tick {s_}"
                                assert np.array_equal(dones[s_], odone.astype(bool)) and np.array_equal(truncs[s_], otrunc.astype(bool)), f"{name} flags tick {s_}"
                                ob.reset(mask=(odone | otrunc).astype(bool))
                                assert np.array_equal(obs[s_], ob.obs_raw_u8()), f"{name} packed={packed} raw obs tick {s_}"
                            if not packed:  # ... then the drop-in API on the same handle (k_sample_philox + k_step of the same instantiation)
                                for s_ in range(8):
                                    a = env.sample_actions().clone()
                                    oa = ob.sample_actions()
                                    assert np.array_equal(np_(a), oa), f"{name} api actions step {s_}"
                                    _, rew, done, trunc, _ = env.step(a)
                                    orew, odone, otrunc, _ = ob.step(oa)
                                    assert np.array_equal(np_(rew).astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"{name} api rewards step {s_}"
                                    assert np.array_equal(np_(done), odone.astype(bool)) and np.array_equal(np_(trunc), otrunc.astype(bool))
                                    ob.reset(mask=(odone | otrunc).astype(bool))
                            env._export(full=True)
                            compare_full_state(env, ob, f"{name} packed={packed}")
                            n_cases += 1
                            del env, ob, traj
                    finally:
                        del CONFIGS[name]
    assert n_cases >= 12


# ------------------------------------------------------------------------------------------------
# observations
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["itg_1v1_nowalls", "base_1v2_j4_14", "base_2v6_j4_14", "tagging_1v4_j5", "itg_1v5_j3"])
def test_observations_match_oracle(pkg, oracle_mod, name):
    B, seed = 777, 3
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False)
    env.reset()
    ob.reset()
    one_imp = env.n_imposters == 1
    comps = ["onehot_pos", "coord_pos", "alive_crew", "walls3x3", "dist_to_imp"]
    if one_imp:
        comps += ["l1_crew", "closest_crew"]
    if env.n_rows == 9:
        comps += ["room_loc"]
    for s in range(40):
        a = env.sample_actions().clone()
        env.step(a)
        orew, odone, otrunc, _ = ob.step(ob.sample_actions())
        ob.reset(mask=(odone | otrunc).astype(bool))
        if s % 8 == 0:
            np.testing.assert_array_equal(np_(env.observe(pkg.ObsConfig("raw"))), ob.obs_raw().astype(np.float32))
            if env.VARIANT != 2:  # the reference featurizers do not run on the tagging env
                got = np_(env.observe(pkg.ObsConfig("flat", comps)))
                want = ob.obs_flat(comps)
                assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), f"{name} flat step {s}"
                i8 = np_(env.observe(pkg.ObsConfig("flat", comps, dtype=torch.int8)))
                np.testing.assert_array_equal(i8.astype(np.float32), want)
            sp, non = env.observe(pkg.ObsConfig("planes"))
            wsp, wnon = ob.obs_planes()
            np.testing.assert_array_equal(np_(sp), wsp)
            np.testing.assert_array_equal(np_(non), wnon)


def test_observations_match_reference_feature_fixtures(pkg):
    """Directly against golden vectors produced by the reference featurizers (states injected)."""
    import glob
    import os

    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "feat_*.npz"))):
        g = load_golden(path)
        meta = g["meta"]
        S = len(g["pos"])
        env = env_from_meta(pkg, meta, S, rng="philox")
        env.reset()
        kw = dict(agent_positions=g["pos"], alive_agents=g["alive"], imposter_mask=g["imp"])
        if env.n_jobs:
            kw.update(job_positions=g["jobpos"], completed_jobs=g["jobdone"])
        env.set_state(**kw)
        np.testing.assert_array_equal(np_(env.observe(pkg.ObsConfig("raw"))), g["raw"].astype(np.float32))
        comps = [c for c in meta["flat"] if c != "scent"]
        got = np_(env.observe(pkg.ObsConfig("flat", comps)))
        want = np.concatenate([g["flat_" + c] for c in comps], axis=1)
        assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), g["name"]
        if "planes_spatial" in g:
            sp, non = env.observe(pkg.ObsConfig("planes"))
            np.testing.assert_array_equal(np_(sp), g["planes_spatial"])
            np.testing.assert_array_equal(np_(non), g["planes_non_spatial"])
            # the reference's generate_featurized_states() contract (model_ready.py:175-216, 291-306)
            pf = pkg.PerspectiveFeaturizer(env)
            pf.fit()
            for i, (psp, pns) in enumerate(pf.generate_featurized_states()):
                np.testing.assert_array_equal(np_(psp)[:, 0], g["persp_spatial"][:, i])
                np.testing.assert_array_equal(np_(pns)[:, 0], g["persp_non_spatial"][:, i])
            gf = pkg.GlobalFeaturizer(env)
            gf.fit()
            views = gf.generate_featurized_states()
            assert len(views) == env.n_agents and views[1][1].shape[-1] == non.shape[-1] + env.n_agents
            assert float(views[1][1][0, 0, non.shape[-1] + 1]) == 1.0
        ff = pkg.FlatFeaturizer(env, comps)
        ff.fit()
        zs, fs = ff.generate_featurized_states()[0]
        assert tuple(zs.shape) == (S, 1, 1) and np.array_equal(np_(fs)[:, 0], want)


def test_featurize_flattened_state_windows_matches_reference_fixtures(pkg, oracle_mod):
    """susnet_featurize: the reference's `featurizer.fit(state_sequence[B, T, S])` on flattened-state rows, fed the
    fixture rows `env.flatten_state` produced (float64, as the trainer's window is) in a [B, T, S] shape and in every
    accepted dtype; outputs vs the reference featurizers' golden vectors."""
    import glob
    import os

    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "feat_*.npz"))):
        g = load_golden(path)
        meta = g["meta"]
        n = len(g["raw"])
        T = 4
        B = n // T
        env = env_from_meta(pkg, meta, 8, rng="philox")  # the handle only supplies the configuration
        env.reset()
        rows64 = torch.from_numpy(g["raw"][:B * T]).reshape(B, T, -1)  # host float64, like train.py:318
        comps = [c for c in meta["flat"] if c != "scent"]
        want = np.concatenate([g["flat_" + c] for c in comps], axis=1)[:B * T].reshape(B, T, -1)
        for dt in (torch.float64, torch.float32, torch.uint8, torch.int32, torch.int64):
            got = np_(env.featurize(rows64.to(dt), pkg.ObsConfig("flat", comps)))
            assert got.shape == want.shape and got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), (g["name"], dt)
        ff = pkg.FlatFeaturizer(env, comps)
        ff.fit(rows64)
        zs, fs = ff.generate_featurized_states()[-1]
        assert tuple(zs.shape) == (B, T, 1) and np.array_equal(np_(fs), want)
        if "scent" in meta["flat"]:  # the real-valued component (torch arithmetic), alone and inside a composite
            sc = g["flat_scent"][:B * T].reshape(B, T, 4)
            f2 = pkg.FlatFeaturizer(env, ["scent"])
            f2.fit(rows64)
            assert np_(f2.generate_featurized_states()[0][1]).view(np.uint32).tolist() == sc.view(np.uint32).tolist(), g["name"]
            mix = [comps[0], "scent"] + comps[1:2]
            f3 = pkg.FlatFeaturizer(env, mix)
            f3.fit(rows64)
            want3 = np.concatenate([(sc if c == "scent" else g["flat_" + c][:B * T].reshape(B, T, -1)) for c in mix], axis=2)
            assert np.array_equal(np_(f3.generate_featurized_states()[0][1]), want3)
            assert int(f3.featurized_shape[1][0]) == want3.shape[-1]
        if "planes_spatial" in g:
            sp, non = env.featurize(rows64.cuda(), pkg.ObsConfig("planes"))
            np.testing.assert_array_equal(np_(sp), g["planes_spatial"][:B * T].reshape(B, T, *g["planes_spatial"].shape[1:]))
            np.testing.assert_array_equal(np_(non), g["planes_non_spatial"][:B * T].reshape(B, T, -1))
            pf = pkg.PerspectiveFeaturizer(env)
            pf.fit(rows64)
            for i, (psp, pns) in enumerate(pf.generate_featurized_states()):
                np.testing.assert_array_equal(np_(psp).reshape(B * T, *psp.shape[2:]), g["persp_spatial"][:B * T, i])
                np.testing.assert_array_equal(np_(pns).reshape(B * T, -1), g["persp_non_spatial"][:B * T, i])
        # ragged tail (not a multiple of the 64-row wave) and a bad row
        tail = torch.from_numpy(g["raw"][:70]).reshape(1, 70, -1)
        got = np_(env.featurize(tail, pkg.ObsConfig("flat", comps)))
        assert np.array_equal(got[0], np.concatenate([g["flat_" + c] for c in comps], axis=1)[:70])
        bad = tail.clone()
        bad[0, 5, 0] = 99.0  # x far outside the grid
        with pytest.raises(IndexError):
            env.featurize(bad, pkg.ObsConfig("flat", comps))
        env.check_errors = False
        got = np_(env.featurize(bad, pkg.ObsConfig("flat", comps)))
        assert not got[0, 5].any() and np.array_equal(got[0, 6], want.reshape(B * T, -1)[6])
        with pytest.raises(IndexError):
            env.poll_errors()


@pytest.mark.parametrize("fixture,comps", [("feat_itg_1v1_nowalls", ["onehot_pos"]),
                                           ("feat_base14_1v2_j4", ["onehot_pos", "alive_crew", "closest_crew"])])
def test_compiled_in_featurize_layouts_match_reference_fixtures(pkg, fixture, comps, monkeypatch):
    """The specialised susnet_featurize kernels (k_featurize_flat<FlatRow<..>>: the two FlatFeaturizer layouts the reference's
    experiments use) on the reference's own flattened states, in every accepted row dtype, against the reference featurizers'
    golden vectors; ragged tail, a bad row, and bit-equality with the generic kernel (a handle created under
    SUSNET_FORCE_GENERIC) on rows from a fused rollout."""
    g = load_golden(f"{GOLDEN_DIR}/{fixture}.npz")
    meta = g["meta"]
    env = env_from_meta(pkg, meta, 8, rng="philox")  # the handle only supplies the configuration
    env.reset()
    n = len(g["raw"])
    want = np.concatenate([g["flat_" + c] for c in comps], axis=1)
    rows64 = torch.from_numpy(g["raw"])
    for dt in (torch.float64, torch.float32, torch.uint8, torch.int32, torch.int64):
        got = np_(env.featurize(rows64.to(dt), pkg.ObsConfig("flat", comps)))
        assert got.shape == want.shape and got.view(np.uint32).tolist() == want.view(np.uint32).tolist(), (fixture, dt)
    assert n % 64 != 0 or np.array_equal(np_(env.featurize(rows64[:70], pkg.ObsConfig("flat", comps))), want[:70])  # ragged tail
    bad = rows64[:70].clone()
    bad[5, 1] = 99.0  # y far outside the grid
    with pytest.raises(IndexError):
        env.featurize(bad, pkg.ObsConfig("flat", comps))
    env.check_errors = False
    got = np_(env.featurize(bad, pkg.ObsConfig("flat", comps)))
    assert not got[5].any() and np.array_equal(got[6], want[6]) and np.array_equal(got[4], want[4])
    with pytest.raises(IndexError):
        env.poll_errors()
    # rows of real rollout states: specialised == generic kernel, bit for bit
    env.check_errors = True
    roll = env_from_meta(pkg, meta, 4096 + 24, rng="philox", auto_reset=True)
    roll.reset()
    rows = roll.rollout(9, obs=pkg.ObsConfig("raw", dtype=torch.uint8))["obs"].reshape(-1, roll.flattened_state_size)
    monkeypatch.setenv("SUSNET_FORCE_GENERIC", "1")
    gen = env_from_meta(pkg, meta, 8, rng="philox")
    assert gen.native_layout().test_overrides & 1
    gen.reset()
    for dt in (torch.uint8, torch.float32):
        assert torch.equal(env.featurize(rows.to(dt), pkg.ObsConfig("flat", comps)), gen.featurize(rows.to(dt), pkg.ObsConfig("flat", comps)))


def test_featurize_matches_oracle_on_tagging_rows(pkg, oracle_mod):
    """Rows with the tagging fields (used, tag_counts, timer_left): raw rows written by the fused rollout of a
    tagging game go back through susnet_featurize and must give the fused planes observation of the same states."""
    name, B, T = "tagging_1v4_j5", 300, 40
    env, ob = make_pair(pkg, oracle_mod, name, B, 11, auto_reset=True, check_errors=True)
    env.reset()
    traj = env.rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8))
    rows = traj["obs"]  # [T, B, S] uint8
    env2, _ = make_pair(pkg, oracle_mod, name, B, 11, auto_reset=True, check_errors=True)
    env2.reset()
    traj2 = env2.rollout(T, obs=pkg.ObsConfig("planes"))
    sp, non = env.featurize(rows, pkg.ObsConfig("planes"))
    assert torch.equal(sp, traj2["obs"]) and torch.equal(non, traj2["obs_non_spatial"])
    flat = env.featurize(rows, pkg.ObsConfig("flat", ["onehot_pos", "alive_crew", "dist_to_imp"]))
    env3, _ = make_pair(pkg, oracle_mod, name, B, 11, auto_reset=True, check_errors=True)
    env3.reset()
    traj3 = env3.rollout(T, obs=pkg.ObsConfig("flat", ["onehot_pos", "alive_crew", "dist_to_imp"]))
    assert torch.equal(flat, traj3["obs"])


@pytest.mark.parametrize("name,comps,B", [("itg_1v1_nowalls", ["onehot_pos"], 1000), ("itg_1v1_nowalls", ["onehot_pos"], 65536 + 40),
                                          ("itg_1v1_walls", ["onehot_pos"], 1000), ("itg_1v1_walls", ["onehot_pos"], 65536 + 40),  # (experiment_1v1.ipynb: one_hot_wall)
                                          ("base_1v2_j4_14", ["onehot_pos", "alive_crew", "closest_crew"], 1000),
                                          ("base_1v2_j4_14", ["onehot_pos", "alive_crew", "closest_crew"], 65536 + 8)])
def test_compiled_in_flat_feature_rollouts_match_the_oracle(pkg, oracle_mod, name, comps, B):
    """The specialised float32 FlatFeaturizer writers of the fused rollout (susnet_flat.h: rows built as bit masks, stored
    cooperatively) for the two layouts the reference's experiments use: every tick's actions / rewards / flags and the feature
    rows of the post-step (post-reset) state against the oracle's restatement of the reference featurizers, at 16 and 64
    environments per wave with a ragged last wave, and across a chunked launch."""
    T, seed = (40 if B < 4096 else 6), 17
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False)
    env.reset()
    ob.reset(threads=0)
    cfg = pkg.ObsConfig("flat", comps)
    F = {"itg_1v1_nowalls": 36, "itg_1v1_walls": 36, "base_1v2_j4_14": 88}[name]
    # big batches: the rows of a sample of envs (the first and the last, ragged, waves and a random draw)
    pick = np.arange(B) if B < 4096 else np.unique(np.concatenate([np.arange(256), np.arange(B - 300, B), np.random.default_rng(3).integers(0, B, 600)]))
    for n in (T, 3):
        env.set_launch_limit(7 * B * F * 4 + 64 if n == T and B < 4096 else 0)  # small batch: 7 ticks per launch
        traj = env.rollout(n, obs=cfg)
        torch.cuda.synchronize()
        assert traj["obs"].shape == (n, B, F) and traj["obs"].dtype == torch.float32
        obs = np_(traj["obs"])
        for s in range(n):
            oa = ob.sample_actions()
            np.testing.assert_array_equal(np_(traj["actions"])[s], oa, err_msg=f"{name} actions tick {s}")
            orew, odone, otrunc, _ = ob.step(oa, threads=0)
            assert np.array_equal(np_(traj["rewards"])[s].astype(np.float64).view(np.uint64), orew.view(np.uint64))
            np.testing.assert_array_equal(np_(traj["done"])[s], odone.astype(bool))
            np.testing.assert_array_equal(np_(traj["truncated"])[s], otrunc.astype(bool))
            ob.reset(mask=(odone | otrunc).astype(bool))
            np.testing.assert_array_equal(obs[s][pick], ob.obs_flat(comps, envs=pick), err_msg=f"{name} flat features tick {s}")
    env._export(full=True)
    compare_full_state(env, ob, f"{name} after flat-feature rollouts")
    # the generic writer (any component list: here with one more component) agrees on the shared columns
    env2, _ = make_pair(pkg, oracle_mod, name, min(B, 2048), seed, auto_reset=True, check_errors=False)
    env3, _ = make_pair(pkg, oracle_mod, name, min(B, 2048), seed, auto_reset=True, check_errors=False)
    env2.reset(); env3.reset()
    a = env2.rollout(12, obs=cfg)["obs"]
    b = env3.rollout(12, obs=pkg.ObsConfig("flat", comps + ["coord_pos"]))["obs"]
    assert torch.equal(a, b[:, :, :F])


def test_small_attribute_mirrors(pkg):
    """crew_idxs (base.py:283), agent_rewards (base.py:369,387), compute_state_dims (base.py:565-579: values recorded
    from the reference for FourRoomEnv(1, 2, 4) and FourRoomEnvWithTagging(1, 4, 5), its 2x2 quirk included)."""
    SF = pkg.StateFields
    env = pkg.BatchedFourRoomEnv(1, 2, 4, batch=8, seed=1)
    env.reset()
    assert [env.compute_state_dims(f).tolist() for f in list(SF)[:4]] == [[[9, 9], [9, 9]], [3], [[9, 9], [9, 9]], [4]]
    with pytest.raises(IndexError):
        env.compute_state_dims(SF.USED_TAGS)
    tag = pkg.BatchedFourRoomEnvWithTagging(1, 4, 5, batch=8, seed=1)
    tag.reset()
    assert [tag.compute_state_dims(f).tolist() for f in SF] == [[[9, 9], [9, 9]], [5], [[9, 9], [9, 9]], [5], [5], [5], [49]]
    imp, crew = np_(env.imposter_idxs), np_(env.crew_idxs)
    assert crew.shape == (8, 2) and all(sorted(list(imp[b]) + list(crew[b])) == [0, 1, 2] for b in range(8))
    _, rew, *_ = env.step(env.sample_actions())
    assert env.agent_rewards is rew or torch.equal(env.agent_rewards, rew)


# ------------------------------------------------------------------------------------------------
# error behaviour (reference: AssertionError base.py:357-362, IndexError base.py:379-382)
# ------------------------------------------------------------------------------------------------
def test_error_types_match_reference(pkg):
    env = pkg.BatchedFourRoomEnv(1, 2, 2, batch=4, shuffle_imposter_index=False)
    env.reset()
    with pytest.raises(AssertionError):
        env.step(torch.zeros(4, 2, dtype=torch.int64))  # wrong number of actions
    bad = torch.zeros(4, 3, dtype=torch.int64)
    bad[2, 1] = 8  # >= action_space.n
    with pytest.raises(AssertionError):
        env.step(bad)
    bad[2, 1] = 6  # crew member has only 6 actions (0..5): IndexError in the reference
    with pytest.raises(IndexError):
        env.step(bad)
    bad[2, 1] = 5
    env.step(bad)  # fine again
    with pytest.raises(AssertionError):
        pkg.BatchedFourRoomEnv(2, 2, 1, batch=1)  # base.py:247-249
    pkg.BatchedImposterTrainingGround(1, 0, 0, -3, 0, 0, batch=1)  # 1v1 allowed (pred_prey.py:75-76)


# ------------------------------------------------------------------------------------------------
# full-size properties at BASELINE.json's batch (size-independent invariants)
# ------------------------------------------------------------------------------------------------
def test_full_batch_invariants(pkg):
    B, T = 65536, 64
    env = pkg.BatchedFourRoomEnv(1, 2, 4, batch=B, grid_size=14, auto_reset=True, seed=123, check_errors=False)
    env.reset()
    grid = torch.as_tensor(env.grid, device=env.device)
    jobs0 = env.job_positions.clone()
    traj = env.rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8))
    env._export(full=True)
    torch.cuda.synchronize()
    obs = traj["obs"].long()  # [T, B, 3A+3J]
    A, J = 3, 4
    xy = obs[..., : 2 * A].reshape(T, B, A, 2)
    assert int(xy.min()) >= 0 and int(xy.max()) < 14
    # agents that MOVED this tick stand on a cell that is free under the transposed lookup (base.py:551)
    moved = (xy[1:] != xy[:-1]).any(-1) & ~(traj["done"][1:] | traj["truncated"][1:])[..., None]  # obs[s] is post-auto-reset
    assert bool(grid[xy[1:][..., 1], xy[1:][..., 0]][moved].all())
    # a move changes exactly one coordinate by one
    d = (xy[1:] - xy[:-1]).abs().sum(-1)
    assert int(d[moved].max()) == 1
    # rewards of a 1v2 default game are integers from the finite reward table
    vals = set(torch.unique(traj["rewards"]).tolist())
    assert vals <= {0.0, -5.0, 5.0, 3.0, -3.0, 10.0, -10.0, -2.0, 13.0, -13.0, 7.0, -7.0, 15.0, -15.0, 5.0, 8.0, -8.0}, vals
    # job cells are pairwise distinct after every reset
    jp = env.job_positions.long()
    code = jp[..., 0] * 16 + jp[..., 1]
    assert bool((code.sort(-1).values.diff(dim=-1) != 0).all())
    life = env.lifetime_totals().tolist()
    ended = int((traj["done"] | traj["truncated"]).sum())
    assert life[0] == ended and life[1] + life[2] + life[3] >= ended
    # determinism: same seed, same trajectory
    env2 = pkg.BatchedFourRoomEnv(1, 2, 4, batch=B, grid_size=14, auto_reset=True, seed=123, check_errors=False)
    env2.reset()
    traj2 = env2.rollout(T)
    assert torch.equal(traj2["actions"], traj["actions"]) and torch.equal(traj2["rewards"], traj["rewards"])
    assert not torch.equal(jobs0, env.job_positions)  # episodes did end and respawn
    # back-to-back launches are ordered on the stream: 4 x 16 ticks == 1 x 64 ticks (the state blob written by one
    # launch is what the next one loads)
    env3 = pkg.BatchedFourRoomEnv(1, 2, 4, batch=B, grid_size=14, auto_reset=True, seed=123, check_errors=False)
    env3.reset()
    bufs = env3.alloc_rollout(16)
    chunks = []
    for _ in range(4):
        env3.rollout_into(16, bufs)
        chunks.append(bufs["rewards"].clone())
    assert torch.equal(torch.cat(chunks), traj["rewards"])
    env3._export(full=True)
    torch.cuda.synchronize()
    assert torch.equal(env3.agent_positions, env.agent_positions) and torch.equal(env3.job_positions, env.job_positions)


@pytest.mark.parametrize("name,B", [("itg_1v1_nowalls", 65536), ("base_1v2_j4_14", 65536), ("base_2v6_j4_14", 32768), ("tagging_1v4_j5", 65536)])
def test_packed_records_are_reproducible_at_bench_sizes(pkg, oracle_mod, name, B):
    """The benchmarked launches (packed records, the bench's batch per GPU) repeated from the same state give the same bytes.
    This is the cheap net for timing-dependent faults of the kind found this round (a store's data registers overwritten before
    the store had read them: a few waves of a 40 000-env launch differed from run to run)."""
    T = 48
    recs = []
    for rep in range(3):
        env, _ = make_pair(pkg, oracle_mod, name, B, 5, auto_reset=True, check_errors=False)
        if env.record_layout() is None:
            pytest.skip("configuration not compiled in")
        env.reset()
        bufs = env.alloc_rollout(T, obs=pkg.ObsConfig("raw", dtype=torch.uint8), packed=True)
        env.rollout_into(T, bufs)
        env.rollout_into(T, bufs)  # (a second launch: starts mid-stream, other group phases)
        torch.cuda.synchronize()
        recs.append(bufs["record"].clone())
        del env, bufs
    assert torch.equal(recs[0], recs[1]) and torch.equal(recs[0], recs[2])


# ------------------------------------------------------------------------------------------------
# policy in the loop (BASELINE config 5): device-side obs -> MLP -> argmax -> step, checked by replaying the
# recorded actions through the oracle
# ------------------------------------------------------------------------------------------------
def test_policy_rollout_matches_oracle_on_recorded_actions(pkg, oracle_mod):
    B, steps, seed = 512, 60, 21
    comps = ["onehot_pos", "alive_crew", "closest_crew"]
    name = "base_1v2_j4_14"
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=False,
                        obs=pkg.ObsConfig("flat", comps))
    env.reset()
    ob.reset()
    model = pkg.policy.reference_imposter_mlp(env, comps, seed=3)
    assert [m.in_features for m in model.model if hasattr(m, "in_features")] == [88, 256, 128, 64, 16]
    runner = pkg.PolicyRollout(env, model, crew_model=None, components=comps)
    for s in range(steps):
        np.testing.assert_array_equal(np_(env.obs), ob.obs_flat(comps), err_msg=f"fused flat obs step {s}")
        a = runner.act().clone()
        # the imposter slot is the MLP's greedy action on the fused observation
        want = model(torch.zeros(B, 1, 1, device=env.device), env.obs).argmax(1)
        got = a[env.imposter_mask]
        assert torch.equal(got, want)
        # the env's sample_actions drew from the Philox stream: mirror the draw in the oracle
        oa = ob.sample_actions()
        imp = ob.export()["imp"].astype(bool)
        oa[imp] = np_(want)
        np.testing.assert_array_equal(np_(a), oa)
        _, rew, done, trunc, _ = env.step(a)
        orew, odone, otrunc, rc = ob.step(oa)
        assert rc == 0
        assert np.array_equal(np_(rew).astype(np.float64).view(np.uint64), orew.view(np.uint64))
        np.testing.assert_array_equal(np_(done), odone.astype(bool))
        ob.reset(mask=(odone | otrunc).astype(bool))


@pytest.mark.parametrize("name", ["base_1v2_j4_14", "base_2v6_j4_14", "tagging_1v4_j5", "itg_1v1_nowalls"])
def test_policy_actions_kernel(pkg, oracle_mod, name):
    """susnet_policy_actions (the acting step of visualize.py:547-562 in one launch): per team the argmax of its Q row -- first
    maximum, like torch.argmax -- placed by the episode's roles; without a crew network the crew's slots hold exactly the draws
    sample_actions() returns; every output dtype."""
    B = 3000
    env, ob = make_pair(pkg, oracle_mod, name, B, 5, auto_reset=True, check_errors=False)
    env.reset()
    ob.reset()
    gen = torch.Generator(device=env.device)
    gen.manual_seed(1)
    for tick in range(4):
        q_imp = torch.randn(B, env.n_imposter_actions, device=env.device, generator=gen)
        q_crew = torch.randn(B, env.n_crew_actions, device=env.device, generator=gen)
        q_imp[::7, 2] = q_imp[::7].max(dim=1).values  # ties: the first maximum wins
        q_imp[::7, 1] = q_imp[::7, 2]
        imp = torch.from_numpy(ob.export()["imp"].astype(bool)).to(env.device)
        sampled = env.sample_actions().to(torch.int64).clone()
        want_net = torch.where(imp, q_imp.argmax(1, keepdim=True), q_crew.argmax(1, keepdim=True))
        want_rand = torch.where(imp, q_imp.argmax(1, keepdim=True), sampled)
        np.testing.assert_array_equal(np_(sampled), ob.sample_actions())
        for dt in (torch.int64, torch.int32, torch.uint8):
            out = torch.zeros(B, env.n_agents, dtype=dt, device=env.device)
            assert torch.equal(env.policy_actions(q_imp, q_crew, out=out).to(torch.int64), want_net), (name, dt)
            assert torch.equal(env.policy_actions(q_imp, None, out=out).to(torch.int64), want_rand), (name, dt)
        a = env.policy_actions(q_imp, None)  # the env's own int64 buffer; step it so that roles / ticks move on
        env.step(a)
        ob.step(np_(a))
        done = ob.export()
        ob.reset(mask=np_(env._done | env._trunc))


@pytest.mark.parametrize("crew_net", [False, True])
@pytest.mark.parametrize("name", ["base_1v2_j4_14", "tagging_1v4_j5", "itg_1v1_nowalls"])
def test_epsilon_greedy_and_dead_mask(pkg, oracle_mod, name, crew_net):
    """The trainer's acting rule (train.py:351-381) inside the kernels that choose the actions: with probability epsilon an agent takes
    its uniformly random role-valid draw (the action stream's, what sample_actions() returns for it) instead of its team's argmax, dead
    agents get index 0.  The explore decisions are word tick * A + i of the exploration stream (Philox key of the handle, counter word 1
    tagged 0x40000000), restated here from the oracle's Philox function; susnet_policy_actions, susnet_policy_step (twin handle) and
    the share of explored actions are checked tick by tick."""
    B, T, seed, eps = 1500, 30, 29, 0.3
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=True)
    twin, _ = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=True)
    env.reset()
    twin.reset()
    ob.reset()
    A = env.n_agents
    gen = torch.Generator(device=env.device)
    gen.manual_seed(8)
    explored = total = 0
    for tick in range(T):
        q_imp = torch.randn(B, env.n_imposter_actions, device=env.device, generator=gen)
        q_crew = torch.randn(B, env.n_crew_actions, device=env.device, generator=gen) if crew_net else None
        st = ob.export()
        imp, alive = st["imp"].astype(bool), st["alive"].astype(bool)
        sampled = ob.sample_actions().astype(np.int64)
        greedy = np.where(imp, np_(q_imp.argmax(1))[:, None], np_(q_crew.argmax(1))[:, None] if crew_net else sampled)
        explore = np.zeros((B, A), dtype=bool)
        for b in range(B):
            for i in range(A):
                idx = tick * A + i
                w = oracle_mod.philox4x32_10([(idx >> 2) & 0xFFFFFFFF, ((idx >> 2) >> 32) | 0x40000000, b, 0], [seed & 0xFFFFFFFF, seed >> 32])[idx & 3]
                explore[b, i] = np.float32(w >> 8) * np.float32(2.0 ** -24) <= np.float32(eps)
        want = np.where(explore, sampled, greedy)
        want_masked = np.where(alive, want, 0)
        assert int(env.tick) == tick
        np.testing.assert_array_equal(np_(env.policy_actions(q_imp, q_crew, epsilon=eps)), want, err_msg=f"{name} tick {tick}")
        a = env.policy_actions(q_imp, q_crew, epsilon=eps, mask_dead=True).clone()
        np.testing.assert_array_equal(np_(a), want_masked, err_msg=f"{name} tick {tick} (dead mask)")
        np.testing.assert_array_equal(np_(env.policy_actions(q_imp, q_crew, mask_dead=True)), np.where(alive, greedy, 0))
        _, r1, d1, t1, _ = env.step(a)
        _, r2, d2, t2, _, a2 = twin.policy_step(q_imp, q_crew, epsilon=eps, mask_dead=True)
        assert torch.equal(a2, a) and torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(d1, d2) and torch.equal(t1, t2), (name, tick)
        orew, odone, otrunc, rc = ob.step(np_(a))
        assert rc == 0 and np.array_equal(np_(d1), odone.astype(bool))
        ob.reset(mask=(odone | otrunc).astype(bool))
        explored += int(explore.sum())
        total += explore.size
    assert abs(explored / total - eps) < 0.02, "the exploration stream is uniform"


@pytest.mark.parametrize("crew_net", [False, True])
@pytest.mark.parametrize("name", ["base_1v2_j4_14", "base_2v6_j4_14", "tagging_1v4_j5", "itg_1v1_nowalls", "itg_1v5_j3"])
def test_policy_step_equals_policy_actions_then_step(pkg, oracle_mod, name, crew_net):
    """susnet_policy_step (argmax per team, the crew's draws and the step in ONE launch) against the two launches it replaces on a twin
    handle with the same seed: actions (every dtype the twin writes), reward bit patterns, done / truncated and the exported state, tick
    after tick through episode ends and in-step resets; compiled-in and generic configurations."""
    B, T = 2000, 40
    e1, ob = make_pair(pkg, oracle_mod, name, B, 17, auto_reset=True, check_errors=True)
    e2, _ = make_pair(pkg, oracle_mod, name, B, 17, auto_reset=True, check_errors=True)
    e1.reset()
    e2.reset()
    gen = torch.Generator(device=e1.device)
    gen.manual_seed(3)
    for tick in range(T):
        q_imp = torch.randn(B, e1.n_imposter_actions, device=e1.device, generator=gen)
        q_crew = torch.randn(B, e1.n_crew_actions, device=e1.device, generator=gen) if crew_net else None
        q_imp[::5, 1] = q_imp[::5].max(dim=1).values  # ties: the first maximum wins in both paths
        a1 = e1.policy_actions(q_imp, q_crew).clone()
        _, r1, d1, t1, _ = e1.step(a1)
        out = torch.zeros(B, e2.n_agents, dtype=(torch.uint8, torch.int32, torch.int64)[tick % 3], device=e2.device)
        _, r2, d2, t2, _, a2 = e2.policy_step(q_imp, q_crew, actions_out=out)
        assert torch.equal(a2.to(torch.int64), a1), (name, tick)
        assert torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(d1, d2) and torch.equal(t1, t2), (name, tick)
        assert torch.equal(e1.agent_positions, e2.agent_positions) and torch.equal(e1.alive_agents, e2.alive_agents), (name, tick)
    assert int(e1.tick) == int(e2.tick) == T


@pytest.mark.parametrize("crew_net", [False, True], ids=["random-crew", "crew-network"])
@pytest.mark.parametrize("eps", [0.0, 0.25])
@pytest.mark.parametrize("name,comps", [("base_1v2_j4_14", ["onehot_pos", "alive_crew", "closest_crew"]), ("itg_1v1_nowalls", ["onehot_pos"]),
                                        ("itg_1v1_walls", ["coord_pos"])])
def test_one_kernel_policy_tick_equals_two_launches(pkg, oracle_mod, name, comps, eps, crew_net):
    """susnet_qnet_policy_step (network, argmax, crew draws, step: ONE kernel) against susnet_qnet_forward + susnet_policy_step on a twin
    handle with the same seed: actions, reward bit patterns, done / truncated, the fused float observation and the exported state, tick
    after tick; B no multiple of the kernel's 256 environments per workgroup; the Q rows it can also emit.  crew-network (round 5): BOTH
    teams by their networks inside the one kernel (visualize.py:547-562) -- the workgroup swaps the LDS image between the two passes --
    against the twin's two network launches + susnet_policy_step."""
    B, T = 3000 + 41, 60
    obs = pkg.ObsConfig("flat", comps)
    e1, _ = make_pair(pkg, oracle_mod, name, B, 23, auto_reset=True, check_errors=True, obs=obs)
    e2, _ = make_pair(pkg, oracle_mod, name, B, 23, auto_reset=True, check_errors=True, obs=obs)
    e1.reset()
    e2.reset()
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(6)
        model = pkg.MLP([e1.obs.shape[-1], 256, 128, 64, 16, e1.n_imposter_actions]).to(e1.device).eval()
        crew = pkg.MLP([e1.obs.shape[-1], 200, 128, 48, 16, e1.n_crew_actions]).to(e1.device).eval() if crew_net else None
    two = pkg.PolicyRollout(e1, model, crew_model=crew, components=comps, epsilon=eps, mask_dead=eps > 0)  # (0.25: the trainer's acting rule)
    one = pkg.PolicyRollout(e2, model, crew_model=crew, components=comps, epsilon=eps, mask_dead=eps > 0)
    assert one.one_kernel_tick and two.one_kernel_tick, "both games are served by the one-kernel tick"
    assert (one.fused_crew is not None) == crew_net
    two.one_kernel_tick = False  # the twin takes the two-launch path
    ends = 0
    for tick in range(T):
        a1, r1, d1, t1 = two.tick()
        a1 = a1.clone()
        a2, r2, d2, t2 = one.tick()
        assert torch.equal(a1, a2), (name, tick)
        assert torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(d1, d2) and torch.equal(t1, t2), (name, tick)
        assert torch.equal(e1.obs, e2.obs), (name, tick)
        assert torch.equal(e1.agent_positions, e2.agent_positions) and torch.equal(e1.alive_agents, e2.alive_agents), (name, tick)
        ends += int(d1.sum()) + int(t1.sum())
    assert int(e1.tick) == int(e2.tick) == T  # (episode ends inside a policy step: test_policy_step_equals_policy_actions_then_step)
    q = torch.empty(B, e2.n_imposter_actions, device=e2.device)
    want = e2.qnet_forward(one.fused_imposter).clone()
    if not crew_net:
        e2.qnet_policy_step(one.fused_imposter, q_out=q)
    else:
        qc = torch.empty(B, e2.n_crew_actions, device=e2.device)
        want_c = e2.qnet_forward(one.fused_crew).clone()
        e2.qnet_policy_step(one.fused_imposter, q_out=q, net_crew=one.fused_crew, q_crew_out=qc)
        assert torch.equal(qc, want_c), "the crew's Q rows"
    assert torch.equal(q, want), "the Q rows the one-kernel tick emits are the network kernel's"


def test_one_kernel_policy_tick_long_horizon_and_full_size(pkg, oracle_mod):
    """The one-kernel tick (susnet_qnet_policy_step) over many episode ends: 400 ticks of epsilon-greedy acting on 384 envs, the
    recorded actions replayed on the oracle -- rewards bit for bit, done / truncated, the fused float observation after every in-step
    reset.  And at the benchmark's size (65 536 envs) against the two-launch tick on a twin handle, 12 ticks, bit for bit."""
    comps = ["onehot_pos", "alive_crew", "closest_crew"]
    name, B, T, seed = "base_1v2_j4_14", 384, 400, 31
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=True, obs=pkg.ObsConfig("flat", comps))
    env.reset()
    ob.reset()
    model = pkg.policy.reference_imposter_mlp(env, comps, seed=9)
    runner = pkg.PolicyRollout(env, model, crew_model=None, components=comps, epsilon=0.3)
    assert runner.one_kernel_tick
    ends = 0
    for tick in range(T):
        a, rew, done, trunc = runner.tick()
        orew, odone, otrunc, rc = ob.step(np_(a))
        assert rc == 0, f"tick {tick}: the oracle refuses an action the kernel chose"
        assert np.array_equal(np_(rew).astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"rewards tick {tick}"
        np.testing.assert_array_equal(np_(done), odone.astype(bool), err_msg=f"done tick {tick}")
        np.testing.assert_array_equal(np_(trunc), otrunc.astype(bool), err_msg=f"truncated tick {tick}")
        ob.reset(mask=(odone | otrunc).astype(bool))
        if tick % 20 == 0 or tick == T - 1:
            np.testing.assert_array_equal(np_(env.obs), ob.obs_flat(comps), err_msg=f"fused flat obs tick {tick}")
        ends += int(odone.sum()) + int(otrunc.sum())
    assert ends > 50, "episodes ended inside the one-kernel tick"
    # full size (the benchmark's 65 536 envs): against the two-launch tick on a twin handle AND against the oracle (OpenMP batch)
    Bf = 65536
    e1, _ = make_pair(pkg, oracle_mod, name, Bf, seed, auto_reset=True, check_errors=False, obs=pkg.ObsConfig("flat", comps))
    e2, obf = make_pair(pkg, oracle_mod, name, Bf, seed, auto_reset=True, check_errors=False, obs=pkg.ObsConfig("flat", comps))
    e1.reset()
    e2.reset()
    obf.reset(threads=0)
    two = pkg.PolicyRollout(e1, model, crew_model=None, components=comps, epsilon=0.1, mask_dead=True)
    one = pkg.PolicyRollout(e2, model, crew_model=None, components=comps, epsilon=0.1, mask_dead=True)
    two.one_kernel_tick = False
    for tick in range(12):
        a1, r1, d1, t1 = two.tick()
        a1 = a1.clone()
        a2, r2, d2, t2 = one.tick()
        assert torch.equal(a1, a2) and torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(d1, d2) and torch.equal(t1, t2), tick
        assert torch.equal(e1.obs, e2.obs), tick
        orew, odone, otrunc, rc = obf.step(np_(a2), threads=0)
        assert rc == 0 and np.array_equal(np_(r2).astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"full size: rewards tick {tick}"
        np.testing.assert_array_equal(np_(d2), odone.astype(bool), err_msg=f"full size: done tick {tick}")
        np.testing.assert_array_equal(np_(t2), otrunc.astype(bool), err_msg=f"full size: truncated tick {tick}")
        obf.reset(mask=(odone | otrunc).astype(bool))
    np.testing.assert_array_equal(np_(e2.obs), obf.obs_flat(comps), err_msg="full size: fused flat obs after 12 ticks")


@pytest.mark.parametrize("name,comps,hidden,slopes,B", [
    ("base_1v2_j4_14", ["onehot_pos", "alive_crew", "closest_crew"], [256, 128, 64, 16], (0.1, 0.3, 0.5, 0.7), 5037),  # BASELINE config 5's network
    ("base_1v2_j4_14", ["onehot_pos", "alive_crew", "closest_crew"], [200, 100, 50, 10], (0.25, 0.25, 0.0, 1.0), 5037),  # widths that need padding
    ("base_1v2_j4_14", ["onehot_pos", "alive_crew", "closest_crew"], [256, 128, 64, 32], (-0.2, 1.5, 0.3, 0.01), 300),  # slopes outside [0, 1]: compare / select PReLU
    ("base_1v2_j4_14", ["onehot_pos", "alive_crew", "closest_crew"], [1, 1, 1, 1], (0.25, 0.25, 0.25, 0.25), 1),        # the narrowest stack, one environment
    ("itg_1v1_nowalls", ["onehot_pos"], [256, 128, 64, 16], (0.1, 0.3, 0.5, 0.7), 5037),                                # notebooks/experiment_1v1.ipynb
    ("itg_1v1_walls", ["onehot_pos"], [256, 128, 64, 16], (0.25, 0.25, 0.25, 0.25), 1000),                              # ... one_hot_wall
    ("itg_1v1_nowalls", ["coord_pos"], [256, 128, 64, 16], (0.1, 0.3, 0.5, 0.7), 5037),                                 # ... no_wall_coord_features
    ("itg_1v1_walls", ["coord_pos"], [256, 128, 64, 16], (0.25, 0.25, 1.5, -0.5), 777),                                 # ... wall_coord_features
])
def test_qnet_forward_matches_torch_mlp(pkg, oracle_mod, name, comps, hidden, slopes, B):
    """susnet_qnet_forward (state words -> Q rows in one kernel: layer 1 as a gather of W1 columns, layers 2..5 on the f32 matrix
    instructions) against the reference-architecture torch MLP (dqn.py:72-108) evaluated on the env's own fused flat observation: values
    to float32 summation-order tolerance, argmax equal except between numerically tied entries.  B is no multiple of the kernel's 256
    environments per workgroup; a few policy ticks in between so that dead agents and fresh episodes occur; both PReLU forms (every
    slope in [0, 1]: max(x, s x); otherwise compare / select)."""
    env, ob = make_pair(pkg, oracle_mod, name, B, 11, auto_reset=True, check_errors=False, obs=pkg.ObsConfig("flat", comps))
    env.reset()
    spatial = torch.zeros(B, 1, 1, device=env.device)
    for which, n_out in (("imposter", env.n_imposter_actions), ("crew", env.n_crew_actions)):
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(4)
            model = pkg.MLP([env.obs.shape[-1]] + hidden + [n_out]).to(env.device).eval()
            with torch.no_grad():  # default PReLU slopes are all 0.25 and default biases small: make every parameter matter
                for i, mod in enumerate(m for m in model.model if isinstance(m, torch.nn.PReLU)):
                    mod.weight.fill_(slopes[i])
                for mod in (m for m in model.model if isinstance(m, torch.nn.Linear)):
                    mod.bias.uniform_(-0.5, 0.5)
        net = pkg.policy.pack_mlp(env, model, comps)
        assert net is not None, "the compiled-in network family serves this stack"
        for tick in range(6):
            with torch.no_grad():
                want = model(spatial, env.obs)
            got = env.qnet_forward(net)
            torch.cuda.synchronize()
            scale = float(want.abs().max())
            assert float((got - want).abs().max()) <= 2e-5 * max(scale, 1e-3), (name, which, tick, float((got - want).abs().max()), scale)
            ga, wa = got.argmax(1), want.argmax(1)
            gap = want.gather(1, wa[:, None]) - want.gather(1, ga[:, None])
            assert bool((gap <= 1e-5 * max(scale, 1e-3)).all())
            env.step(env.sample_actions())
    assert pkg.policy.pack_mlp(env, pkg.MLP([env.obs.shape[-1], 64, 7]).to(env.device), comps) is None  # another depth: torch serves it


@pytest.mark.parametrize("tag", ["nowalls", "walls"])
def test_qnet_coordinate_layout_matches_the_reference_mlp(pkg, tag):
    """The coordinate feature layout (CoordinateAgentPositionsFeaturizer, component.py:384-403: two of the four live experiments of
    notebooks/experiment_1v1.ipynb) through the Q-network kernel, against the REFERENCE's own objects: tests/golden/model_mlp_coord_1v1_*.npz
    holds states of a 1v1 rollout on either map (24 of them with a dead crew member: its coordinates stay in the row), the reference
    featurizer's rows and the Q rows of two reference `MLP`s.  The states are imported, one per environment; the fused flat observation must
    equal the featurizer's rows exactly and susnet_qnet_forward the Q rows to float32 summation-order tolerance (layer 1 gathers k x W1
    columns multiplied on the host), with the same argmax wherever the reference's margin is not a numerical tie."""
    g = load_golden(f"{GOLDEN_DIR}/model_mlp_coord_1v1_{tag}.npz")
    meta = g["meta"]
    rows = g["rows"].astype(np.int64)
    n = len(rows)
    comps = meta["components"]
    env = env_from_meta(pkg, meta, n, rng="philox", obs=pkg.ObsConfig("flat", comps), check_errors=False)
    env.reset()
    env.set_state(agent_positions=rows[:, :4].reshape(n, 2, 2), alive_agents=rows[:, 4:6])
    feats = env.observe(pkg.ObsConfig("flat", comps))
    np.testing.assert_array_equal(np_(feats), g["features"], err_msg="flat coordinate features")
    assert int((rows[:, 5] == 0).sum()) >= 20
    for team in ("imposter", "crew"):
        model = pkg.policy.MLP(meta["dims"][team])
        model.load_state_dict({k[len(team) + 2:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(team + "::")})
        model = model.to(env.device).eval()
        net = pkg.policy.pack_mlp(env, model, comps)
        assert net is not None, "the Q-network kernel serves the coordinate layout on the 1v1 game"
        got = np_(env.qnet_forward(net))
        want = g["q_" + team]
        scale = float(np.abs(want).max())
        assert float(np.abs(got - want).max()) <= 2e-5 * max(scale, 1e-3), (team, float(np.abs(got - want).max()), scale)
        top2 = np.sort(want, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4 * max(scale, 1e-3)
        np.testing.assert_array_equal(got.argmax(1)[clear], want.argmax(1)[clear], err_msg=team + " argmax")
    # the one-kernel policy tick serves the layout too (network, argmax, the crew's draws and the step in one launch)
    model = pkg.policy.MLP(meta["dims"]["imposter"]).to(env.device).eval()
    pol = pkg.PolicyRollout(env, model, None, components=comps)
    assert pol.one_kernel_tick


def test_captured_policy_tick_replays_against_the_oracle(pkg, oracle_mod):
    """PolicyRollout.capture: the whole policy tick as a hipGraph.  Replays are checked like the eager loop above -- fused flat
    observation, MLP argmax in the imposter slot, Philox crew draws mirrored in the oracle, rewards / done bit for bit -- for the
    two eager warm-up ticks and three replays of a 7-tick graph (21 ticks: episode ends and in-step resets included)."""
    B, seed, n = 512, 21, 7
    comps = ["onehot_pos", "alive_crew", "closest_crew"]
    env, ob = make_pair(pkg, oracle_mod, "base_1v2_j4_14", B, seed, auto_reset=True, check_errors=False, export_state=False,
                        obs=pkg.ObsConfig("flat", comps))
    env.reset()
    ob.reset()
    model = pkg.policy.reference_imposter_mlp(env, comps, seed=3)
    runner = pkg.PolicyRollout(env, model, crew_model=None, components=comps)
    spatial = torch.zeros(B, 1, 1, device=env.device)

    def check_tick(obs_before, a, rew, done, label):
        np.testing.assert_array_equal(np_(obs_before), ob.obs_flat(comps), err_msg=f"fused flat obs, {label}")
        q = model(spatial, obs_before)  # the torch module; the runner's Q rows come from susnet_qnet_forward (float32, another
        want = q.argmax(1)              # summation order): its argmax may differ from torch's only between numerically tied rows
        oa = ob.sample_actions()
        imp = ob.export()["imp"].astype(bool)
        got = a[torch.from_numpy(imp).to(a.device)]
        gap = q.gather(1, want[:, None]) - q.gather(1, got[:, None])
        assert bool((gap <= 1e-5 * q.abs().max()).all()), f"imposter actions are the network's argmax, {label}"
        oa[imp] = np_(got)
        np.testing.assert_array_equal(np_(a), oa, err_msg=f"actions, {label}")
        orew, odone, otrunc, rc = ob.step(oa)
        assert rc == 0
        assert np.array_equal(np_(rew).astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"rewards, {label}"
        np.testing.assert_array_equal(np_(done), odone.astype(bool), err_msg=f"done, {label}")
        ob.reset(mask=(odone | otrunc).astype(bool))

    graph, out = runner.capture(n, record=True)
    torch.cuda.synchronize()
    # the capture's eager warm-up ticks advanced the rollout: they wrote slots 0 and 1 of the record
    for k in range(runner.captured_warmup_ticks):
        check_tick(out["obs_before"][k], out["actions"][k], out["rewards"][k], out["done"][k], f"warm-up tick {k}")
    ends = 0
    for rep in range(3):
        graph.replay()
        torch.cuda.synchronize()
        for k in range(n):
            check_tick(out["obs_before"][k], out["actions"][k], out["rewards"][k], out["done"][k], f"replay {rep} tick {k}")
        ends += int(out["done"].sum()) + int(out["truncated"].sum())
    assert int(env.tick) == runner.captured_warmup_ticks + 3 * n


def test_windowed_spatialdqn_policy_rollout(pkg, oracle_mod):
    """SpatialDQN (CNN + RNN over a T = 3 window of flattened states) acting for every agent through the Perspective
    featurizer, epsilon-greedy: the window follows np.roll / refill-on-reset (train.py:318-322, 388-389, 452-457) with
    the oracle supplying the expected raw rows, the greedy indices are recomputed from the window, and the env's
    rewards / dones for the recorded actions equal the oracle's."""
    B, steps, seed, T = 384, 70, 13, 3
    name = "base_1v2_j4_14"
    env, ob = make_pair(pkg, oracle_mod, name, B, seed, auto_reset=True, check_errors=True,
                        obs=pkg.ObsConfig("raw", dtype=torch.uint8))
    A, J, N = env.n_agents, env.n_jobs, env.n_rows
    torch.manual_seed(5)
    mk = lambda n_act: pkg.SpatialDQN(input_image_size=N, non_spatial_input_size=A + J, n_channels=[A + 2, 6, 8], strides=[1, 1],
                                      paddings=[1, 1], kernel_size=[3, 3], dilations=[1, 1], rnn_layers=1, rnn_hidden_dim=16,
                                      rnn_dropout=0.0, mlp_hidden_layer_dims=[12], n_actions=n_act).to(env.device).eval()
    imp_net, crew_net = mk(env.n_imposter_actions), mk(env.n_crew_actions)
    feat = pkg.PerspectiveFeaturizer(env)
    runner = pkg.WindowedPolicyRollout(env, feat, imp_net, crew_net, sequence_length=T, epsilon=0.25, mask_dead=True, seed=1)
    win = runner.reset()
    ob.reset()
    first = ob.obs_raw_u8()
    want_win = np.repeat(first[:, None, :], T, axis=1)
    np.testing.assert_array_equal(np_(win), want_win)
    explored = 0
    for s in range(steps):
        # greedy indices recomputed from the window the runner holds
        feat.fit(runner.window)
        imp = env.imposter_mask.clone()  # (a live view: refreshed in place by the next step)
        greedy = torch.stack([torch.where(imp[:, i], imp_net(sp, ns).argmax(1), crew_net(sp, ns).argmax(1))
                              for i, (sp, ns) in enumerate(feat.generate_featurized_states())], dim=1)
        alive = torch.from_numpy(ob.export()["alive"].astype(np.int64)).to(env.device)
        a, rew, done, trunc = runner.step()
        a = a.clone()
        assert bool(((a == 0) | (alive == 1)).all()), "dead agents get index 0 (train.py:353-357)"
        limit = torch.where(imp, env.n_imposter_actions, env.n_crew_actions)
        assert bool((a < limit).all()), "role-valid indices"
        same = (a == greedy * alive)
        explored += int((~same).sum())
        assert float(same.float().mean()) > 0.6, "mostly greedy at epsilon 0.25"
        orew, odone, otrunc, rc = ob.step(np_(a).astype(np.int64))
        assert rc == 0
        assert np.array_equal(np_(rew).astype(np.float64).view(np.uint64), orew.view(np.uint64)), f"rewards step {s}"
        np.testing.assert_array_equal(np_(done), odone.astype(bool))
        ended = (odone | otrunc).astype(bool)
        ob.reset(mask=ended)
        nxt = ob.obs_raw_u8()
        want_win = np.roll(want_win, -1, axis=1)
        want_win[:, -1] = nxt
        want_win[ended] = nxt[ended][:, None, :]
        np.testing.assert_array_equal(np_(runner.window), want_win, err_msg=f"window step {s}")
    assert explored > 0


def replay_names():
    import glob
    import os

    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "replay_*.npz")))


def _replay_feeds():
    # (susnet_ring_append reads packed records in place where the handle stores them whole: the 1v1 kernels)
    return [(n, f) for n in replay_names() for f in ("tensors",) + (("records", "compact") if n.startswith("replay_itg_1v1") else ())]


@pytest.mark.parametrize("tile", [None, "8"], ids=["row-kernel", "tile-kernel"])
@pytest.mark.parametrize("name,feed", _replay_feeds())
def test_native_replay_ring_matches_reference_populate(pkg, name, feed, tile, monkeypatch):
    """susnet_ring_append fed by the fused rollout against the REFERENCE's own `ReplayBuffer.populate` tensors
    (tests/golden/generate_replay.py: unmodified src/replay_memory.py on the unmodified env after np.random.seed): one env,
    numpy's words as the tape, several launches (the window is carried across them; one fixture wraps the ring)."""
    g = load_golden(f"{GOLDEN_DIR}/{name}.npz")
    meta = g["meta"]
    T, max_size, num_steps = meta["trajectory_size"], meta["max_size"], meta["num_steps"]
    if tile is not None:
        monkeypatch.setenv("SUSNET_RING_TILE", tile)  # (read at susnet_create)
    env = env_from_meta(pkg, meta, 1, rng="numpy", tape_words=1 << 16, auto_reset=True, check_errors=False)
    env._reseed([meta["seed"]])
    buf = pkg.DeviceReplayBuffer(max_size, meta["state_size"], T, meta["n_agents"], meta["n_imposters"], device=env.device)
    if feed != "tensors":  # susnet_ring_append reading the rollout's packed records in place (whole records: the 1v1 kernels)
        lay = env.record_layout(True if feed == "records" else "compact")
        if lay is None or lay.planar:
            pytest.skip("whole packed records of this format: the 1v1 kernels only")
    assert buf.populate_fused(env, num_steps, ticks_per_launch=256, packed={"tensors": False, "records": True, "compact": "compact"}[feed]) == num_steps
    env.poll_errors()
    assert (buf.idx, buf.size) == (meta["idx"], meta["size"])
    n = buf.size
    np.testing.assert_array_equal(np_(buf.states[:n]), g["states"].astype(np.float32), err_msg="states")
    np.testing.assert_array_equal(np_(buf.next_states[:n]), g["next_states"].astype(np.float32), err_msg="next_states")
    np.testing.assert_array_equal(np_(buf.actions[:n]), g["actions"].astype(np.int64), err_msg="actions")
    assert np_(buf.rewards[:n]).view(np.uint32).tolist() == g["rewards"].view(np.uint32).tolist(), "rewards"
    np.testing.assert_array_equal(np_(buf.dones[:n]), g["dones"].astype(bool), err_msg="dones")
    # the reference keeps numpy's draw order of the imposter indices, the ring stores them ascending (sus-net_amd/replay.py): the
    # rows are compared as they are wherever the order is defined by the data (one imposter, or no shuffle), as sets otherwise
    if meta["n_imposters"] == 1 or not meta["kwargs"].get("shuffle_imposter_index", True):
        np.testing.assert_array_equal(np_(buf.imposters[:n]), g["imposters"], err_msg="imposters")
    else:
        np.testing.assert_array_equal(np_(buf.imposters[:n]), np.sort(g["imposters"], axis=1), err_msg="imposters (as sets)")


def test_policy_rollout_follows_weight_updates(pkg):
    """The packed device image of a Q-network is refreshed when the module's parameters change (optimizer.step / load_state_dict between
    ticks: train.py:402-416): the eager paths check the parameters' versions, `refresh_weights()` re-packs in place (same device buffer: a
    captured graph keeps reading it)."""
    comps = ["onehot_pos", "alive_crew", "closest_crew"]
    env = pkg.BatchedFourRoomEnv(1, 2, 4, grid_size=14, batch=256, auto_reset=True, seed=5, shuffle_imposter_index=False, check_errors=False,
                                 obs=pkg.ObsConfig("flat", comps))
    model = pkg.policy.reference_imposter_mlp(env, comps, seed=1)
    pol = pkg.PolicyRollout(env, model, None, components=comps)
    env.reset()
    ptr = pol.fused_imposter.packed.data_ptr()
    q0 = pol.q_rows()[0].clone()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(0.5).add_(0.01)  # (what an optimizer step does: in-place updates)
    q1 = pol.q_rows()[0].clone()
    with torch.no_grad():
        want = model(torch.zeros(env.batch, 1, 1, device=env.device), env.obs)
    assert not torch.allclose(q0, q1) and torch.allclose(q1, want, rtol=0, atol=2e-5 * float(want.abs().max()))
    assert pol.fused_imposter.packed.data_ptr() == ptr, "re-packed in place"
    model.load_state_dict({k: v * 2 for k, v in model.state_dict().items()})
    assert pol.refresh_weights(force=False) and not pol.refresh_weights(force=False)
    with torch.no_grad():
        want2 = model(torch.zeros(env.batch, 1, 1, device=env.device), env.obs)
    torch.testing.assert_close(pol.q_rows()[0], want2, rtol=0, atol=2e-5 * float(want2.abs().max()))


def collect_names():
    import glob
    import os

    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "collect_*.npz")))


def _collect_models(pkg, g, device):
    meta = g["meta"]
    out = []
    for team in ("imposter", "crew"):
        m = pkg.policy.MLP(meta[f"{team}_dims"])
        m.load_state_dict({k[len(team) + 2:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(team + "::")})
        out.append(m.to(device).eval())
    return out


@pytest.mark.parametrize("name", collect_names())
def test_policy_driven_collection_matches_the_reference_trainer_loop(pkg, name):
    """`DeviceReplayBuffer.collect` -- the policy tick (Q-network kernels + susnet_policy_step writing the replay feed) and
    susnet_ring_append -- against the ring the REFERENCE's objects produce in the trainer's collection loop (tests/golden/
    generate_collect.py: train.py:316-322, 345-399, 419-449 restated around the unmodified env, FlatFeaturizer, MLPs and ReplayBuffer;
    greedy, dead agents get index 0): one env, numpy's words as the tape, the stored network parameters; the actions the kernels chose
    equal the recorded ones because the ring's action rows do."""
    g = load_golden(f"{GOLDEN_DIR}/{name}.npz")
    meta = g["meta"]
    T, max_size, num_steps = meta["trajectory_size"], meta["max_size"], meta["num_steps"]
    comps = meta["components"]
    if "kills" in name:  # (round 5) episodes that END: `done` rows whose next_state is the terminal state, then a fresh episode's first row
        assert int(g["dones"].sum()) >= 15, "the fixture's imposter network chases and kills (generate_collect.py chase_parameters)"
    env = env_from_meta(pkg, meta, 1, rng="numpy", tape_words=1 << 16, auto_reset=True, check_errors=False, obs=pkg.ObsConfig("flat", comps))
    env._reseed([meta["seed"]])
    imp, crew = _collect_models(pkg, g, env.device)
    policy = pkg.PolicyRollout(env, imp, crew, components=comps, mask_dead=True)
    assert policy.fused_imposter is not None and policy.fused_crew is not None, "both reference MLPs are served by the Q-network kernel"
    buf = pkg.DeviceReplayBuffer(max_size, meta["state_size"], T, meta["n_agents"], meta["n_imposters"], device=env.device)
    env.reset()
    for part in (37, num_steps - 37):  # two calls: the sequence window carries over
        assert buf.collect(env, policy, part, epsilon=0.0, mask_dead=True, ticks_per_append=50) == part
    torch.cuda.synchronize()
    env.poll_errors()
    assert (buf.idx, buf.size) == (meta["idx"], meta["size"])
    n = buf.size
    np.testing.assert_array_equal(np_(buf.actions[:n]), g["actions"].astype(np.int64), err_msg="actions (the networks' greedy choices)")
    np.testing.assert_array_equal(np_(buf.states[:n]), g["states"].astype(np.float32), err_msg="states")
    np.testing.assert_array_equal(np_(buf.next_states[:n]), g["next_states"].astype(np.float32), err_msg="next_states")
    assert np_(buf.rewards[:n]).view(np.uint32).tolist() == g["rewards"].view(np.uint32).tolist(), "rewards"
    np.testing.assert_array_equal(np_(buf.dones[:n]), g["dones"].astype(bool), err_msg="dones")
    np.testing.assert_array_equal(np_(buf.imposters[:n]), g["imposters"], err_msg="imposters")


@pytest.mark.parametrize("one_launch", [True, False], ids=["block-in-one-launch", "launch-per-tick"])
def test_one_kernel_collection_matches_the_step_by_step_policy_loop(pkg, one_launch):
    """Production stream, random crew, epsilon-greedy: `collect` (ONE kernel per tick writing the replay feed) against a twin env driven
    tick by tick through `PolicyRollout.tick` with the window / ring bookkeeping done in torch (`add_batch`): same transitions, same
    ring -- episode ends, terminal next-states and truncations included."""
    B, T, n_ticks, comps = 384, 2, 90, ["onehot_pos", "alive_crew", "closest_crew"]
    mk = lambda: pkg.BatchedFourRoomEnv(1, 2, 4, grid_size=14, batch=B, auto_reset=True, seed=77, max_time_steps=20, shuffle_imposter_index=False,
                                        check_errors=False, export_state=False, obs=pkg.ObsConfig("flat", comps))
    env, twin = mk(), mk()
    model = pkg.policy.reference_imposter_mlp(env, comps, seed=3)
    S, A = env.flattened_state_size, env.n_agents
    buf = pkg.DeviceReplayBuffer(B * n_ticks, S, T, A, 1, device=env.device)
    ref = pkg.DeviceReplayBuffer(B * n_ticks, S, T, A, 1, device=env.device)
    pol = pkg.PolicyRollout(env, model, None, components=comps, epsilon=0.25, mask_dead=True)
    assert pol.one_kernel_tick
    env.reset()
    assert buf.collect(env, pol, n_ticks, epsilon=0.25, mask_dead=True, ticks_per_append=32, one_launch_per_block=one_launch) == B * n_ticks
    # the twin: the tested tick (susnet_qnet_policy_step through PolicyRollout) + torch bookkeeping with the reference's window rules
    tp = pkg.PolicyRollout(twin, model, None, components=comps, epsilon=0.25, mask_dead=True)
    raw8 = pkg.ObsConfig("raw", dtype=torch.uint8)
    twin.reset()
    window = twin.observe(raw8).float().unsqueeze(1).repeat(1, T, 1)
    imposters = torch.zeros(B, 1, dtype=torch.int16, device=twin.device)
    ends = 0
    for _ in range(n_ticks):
        before = twin.observe(raw8)
        a, rew, done, trunc = tp.tick()
        after = twin.observe(raw8).float()  # (after the auto-reset where the episode ended)
        ended = done | trunc
        ends += int(ended.sum())
        nxt = torch.roll(window, shifts=-1, dims=1)
        nxt[:, -1] = after
        keep = ~ended
        # a transition that ended its episode: its next window ends with the TERMINAL state, which the twin no longer holds -- compare those
        # rows' last slot through the collecting env's own feed instead (checked separately below); everything else row by row
        ref.add_batch(window, a, rew, nxt, done, imposters)
        window = torch.where(ended.view(-1, 1, 1), after.unsqueeze(1).expand(-1, T, -1), nxt)
    torch.cuda.synchronize()
    assert ends > 100, "the run must cross many episode ends"
    n = B * n_ticks
    np.testing.assert_array_equal(np_(buf.actions[:n]), np_(ref.actions[:n]))
    assert np_(buf.rewards[:n]).view(np.uint32).tolist() == np_(ref.rewards[:n]).view(np.uint32).tolist()
    np.testing.assert_array_equal(np_(buf.dones[:n]), np_(ref.dones[:n]))
    np.testing.assert_array_equal(np_(buf.states[:n]), np_(ref.states[:n]))
    # next_states: equal wherever the episode went on; where it ended the ring holds the true terminal state, which differs from the
    # fresh state the twin's bookkeeping stored -- and must continue the previous state by one move per agent at most
    ended_rows = np_((buf.next_states[:n, -1] != ref.next_states[:n, -1]).any(dim=1))
    np.testing.assert_array_equal(np_(buf.next_states[:n, :-1]), np_(ref.next_states[:n, :-1]))
    prev, term = np_(buf.states[:n, -1])[ended_rows], np_(buf.next_states[:n, -1])[ended_rows]
    assert ended_rows.sum() > 0 and (np.abs(term[:, :2 * A] - prev[:, :2 * A]).reshape(-1, A, 2).sum(-1) <= 1).all()
    np.testing.assert_array_equal(term[:, 3 * A:3 * A + 8], prev[:, 3 * A:3 * A + 8], err_msg="job cells stay within an episode")


@pytest.mark.parametrize("crew_net", [False, True], ids=["random-crew", "crew-network"])
@pytest.mark.parametrize("name,comps,B", [("base_1v2_j4_14", ["onehot_pos", "alive_crew", "closest_crew"], 16 * 61), ("itg_1v1_nowalls", ["onehot_pos"], 16 * 37),
                                          ("itg_1v1_walls", ["coord_pos"], 16 * 37)])
def test_policy_block_in_one_launch_equals_one_launch_per_tick(pkg, oracle_mod, name, comps, B, crew_net):
    """susnet_qnet_policy_rollout (the policy tick looped INSIDE the kernel, the wave re-reading the state it stored) against
    susnet_qnet_policy_step called once per tick on a twin: every feed array of every tick, the Q rows, and the state afterwards.
    B is not a multiple of the 256 environments of a workgroup (so some waves of the last workgroup hold no environment: with the crew's
    network they still take part in the image swaps), and the run crosses episode ends."""
    n = 40
    mk = lambda: make_pair(pkg, oracle_mod, name, B, 5, auto_reset=True, check_errors=False, max_time_steps=12, obs=pkg.ObsConfig("flat", comps))[0]
    env, twin = mk(), mk()
    model = pkg.policy.reference_imposter_mlp(env, comps, seed=9)
    crew = None
    if crew_net:
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(10)
            crew = pkg.MLP([env.obs.shape[-1], 96, 64, 32, 16, env.n_crew_actions]).to(env.device).eval()
    pol, tp = (pkg.PolicyRollout(e, model, crew, components=comps, epsilon=0.2, mask_dead=True) for e in (env, twin))
    if not env.supports_qnet_policy_step(pol.fused_imposter, pol.fused_crew, 0.2):
        pytest.skip("no one-kernel tick for this game")
    env.reset(), twin.reset()
    nq = pol.fused_imposter.dims[-1]
    fa, fb = env.alloc_feed(n), twin.alloc_feed(n)
    qa = torch.zeros(n, B, nq, device=env.device)
    qb = torch.zeros(n, B, nq, device=env.device)
    env.policy_rollout_into(fa, n, pol.fused_imposter, epsilon=0.2, mask_dead=True, q_out=qa, net_crew=pol.fused_crew)
    for k in range(n):
        twin.policy_tick_into(fb, k, tp.fused_imposter, net_crew=tp.fused_crew, epsilon=0.2, mask_dead=True, q_out=qb[k])
    torch.cuda.synchronize()
    for key in ("actions", "rewards", "done", "truncated", "obs", "term_obs", "roles"):
        a, b = fa[key][:n].contiguous(), fb[key][:n].contiguous()
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), key
    assert torch.equal(qa.view(torch.int32), qb.view(torch.int32)), "Q rows"
    assert int((fa["done"][:n] | fa["truncated"][:n]).sum()) > B // 8, "the block must cross episode ends"
    raw8 = pkg.ObsConfig("raw", dtype=torch.uint8)
    assert torch.equal(env.observe(raw8), twin.observe(raw8)) and torch.equal(env.episode_index(), twin.episode_index())
    # and a second block continues where the first stopped (tick counter, episode counters)
    env.policy_rollout_into(fa, 7, pol.fused_imposter, epsilon=0.2, mask_dead=True, net_crew=pol.fused_crew)
    for k in range(7):
        twin.policy_tick_into(fb, k, tp.fused_imposter, net_crew=tp.fused_crew, epsilon=0.2, mask_dead=True)
    torch.cuda.synchronize()
    assert torch.equal(fa["actions"][:7], fb["actions"][:7]) and torch.equal(fa["obs"][:7], fb["obs"][:7])


def test_policy_block_rejects_unaligned_observation_slots(pkg, oracle_mod):
    """susnet_qnet_policy_rollout writes the raw observation of every tick in 16-byte pieces: a batch whose slots would not start on
    16-byte boundaries is refused with SUSNET_E_INVALID (one tick per call and a feed without `obs` stay available)."""
    import ctypes as C

    comps = ["onehot_pos", "alive_crew", "closest_crew"]
    env = make_pair(pkg, oracle_mod, "base_1v2_j4_14", 5, 3, auto_reset=True, check_errors=False, obs=pkg.ObsConfig("flat", comps))[0]
    env.reset()
    pol = pkg.PolicyRollout(env, pkg.policy.reference_imposter_mlp(env, comps, seed=1), None, components=comps)
    L = pkg._lib if hasattr(pkg, "_lib") else __import__("importlib").import_module("sus-net_amd._lib")
    net, feed = pol.fused_imposter, env.alloc_feed(4)
    io = L.FeedIO()
    io.obs = feed["obs"].data_ptr()
    args = (env._h, net.components, len(net.components), net.cdims, len(net.dims), net.packed.data_ptr(), env._policy_opts(0.0, False), C.byref(io))
    assert env.lib.susnet_qnet_policy_rollout(*args, 4, env._stream()) == L.E_INVALID and b"16" in env.lib.susnet_last_error()
    assert env.lib.susnet_qnet_policy_rollout(*args, 1, env._stream()) == 0  # one tick: slot 0 only
    env.policy_block(3, net)  # no feed at all
    torch.cuda.synchronize()
    assert int(env.tick) == 4


@pytest.mark.parametrize("B", [16 * 53, 65536], ids=["848-envs", "bench-size"])
def test_policy_rollout_run_in_blocks_equals_tick_by_tick(pkg, oracle_mod, B):
    """`PolicyRollout.run` with the ticks of a block in ONE launch (env.policy_block: susnet_qnet_policy_rollout without a feed) against one
    launch per tick on a twin: the state, the refreshed fused observation, the step counter and the episode metrics afterwards.  At the
    bench's batch every SIMD holds a wave that re-reads, tick after tick, what it stored the tick before (wavefront-scope fences only)."""
    comps = ["onehot_pos", "alive_crew", "closest_crew"]
    mk = lambda: make_pair(pkg, oracle_mod, "base_1v2_j4_14", B, 9, auto_reset=True, check_errors=False, max_time_steps=15,
                           obs=pkg.ObsConfig("flat", comps))[0]
    env, twin = mk(), mk()
    model = pkg.policy.reference_imposter_mlp(env, comps, seed=4)
    a, b = (pkg.PolicyRollout(e, model, None, components=comps, epsilon=0.15, mask_dead=True) for e in (env, twin))
    assert a.one_kernel_tick
    env.reset(), twin.reset()
    a.run(45, block_ticks=16)  # launches of 16, 16, 13 ticks
    b.run(45, block_ticks=0)
    torch.cuda.synchronize()
    raw8 = pkg.ObsConfig("raw", dtype=torch.uint8)
    assert torch.equal(env.observe(raw8), twin.observe(raw8)) and torch.equal(env.obs, twin.obs)
    assert int(env.tick) == int(twin.tick) == 45 and torch.equal(env.episode_index(), twin.episode_index())
    assert np.array_equal(np_(env.lifetime_totals()), np_(twin.lifetime_totals())) and int(env.lifetime_totals()[0]) > B


@pytest.mark.parametrize("game,T,tile", [("base_1v2", 3, None), ("base_1v2", 3, "tile=8"),
                                         ("base_1v2", 3, "tile=16"), ("base_1v2", 2, "tile=32"), ("base_2v6", 8, "tile=8"), ("base_2v6", 8, None),
                                         ("tagging_1v4", 14, None), ("base_2v6", 29, None)])
def test_native_replay_ring_batched_matches_a_host_rebuild(pkg, game, T, tile, monkeypatch):
    """Many envs, odd launch lengths, a ring smaller than the run: every row the ring holds equals what replaying the
    trajectory on the host with ReplayBuffer.populate's rules gives (window roll, first state repeated after a reset, true
    terminal next state, `done` only), at position (rows added so far) % max_size.  The long windows (14 x 41 bytes, 29 x 36
    bytes per row) do not fit 64 rows into the kernel's LDS images: a wave then takes 32 / 16 rows (RingArgs::rows_per_wave).  Behind the
    SUSNET_RING_TILE test hook: the (ticks x envs) tile kernel for windows of up to 8 states (8 x 8, 4 x 16, 2 x 32)."""
    if tile is not None:
        monkeypatch.setenv("SUSNET_RING_TILE", tile.split("=")[1])  # (read at susnet_create)
    B, max_size = 300, 4000
    mk = {"base_1v2": lambda: pkg.BatchedFourRoomEnv(1, 2, 4, batch=B, auto_reset=True, seed=31, max_time_steps=25, check_errors=False),
          "tagging_1v4": lambda: pkg.BatchedFourRoomEnvWithTagging(1, 4, 5, batch=B, auto_reset=True, seed=31, max_time_steps=25, check_errors=False),
          "base_2v6": lambda: pkg.BatchedFourRoomEnv(2, 6, 4, batch=B, auto_reset=True, seed=31, max_time_steps=25, check_errors=False)}[game]
    env = mk()
    S, A = env.flattened_state_size, env.n_agents
    buf = pkg.DeviceReplayBuffer(max_size, S, T, A, env.n_imposters, device=env.device)
    # host model: same env / seed stepped through the fused rollout in the same launch pattern
    env2 = mk()
    raw8 = pkg.ObsConfig("raw", dtype=torch.uint8)
    env2.reset()
    first = np_(env2.observe(raw8))
    window = np.repeat(first[:, None, :], T, axis=1).astype(np.float32)
    rows = []
    pattern = [5, 1, 2, 7, 9]
    for n in pattern:
        bufs = env2.alloc_rollout(n, obs=raw8, replay_feed=True)
        env2.rollout_into(n, bufs)
        torch.cuda.synchronize()
        acts, rews, dn, tr, obs, tob, roles = (np_(bufs[k]) for k in ("actions", "rewards", "done", "truncated", "obs", "term_obs", "roles"))
        for t in range(n):
            ended = dn[t] | tr[t]
            nxt_state = np.where(ended[:, None], tob[t], obs[t]).astype(np.float32)
            nxt = np.roll(window, -1, axis=1)
            nxt[:, -1] = nxt_state
            imp = np.array([[i for i in range(A) if (int(roles[t, b]) >> i) & 1] for b in range(B)])
            for b in range(B):
                rows.append((window[b].copy(), acts[t, b].astype(np.int64), rews[t, b], nxt[b].copy(), bool(dn[t, b]), imp[b]))
            fresh = np.repeat(obs[t][:, None, :], T, axis=1).astype(np.float32)
            window = np.where(ended[:, None, None], fresh, nxt)
    added = buf.populate_fused(env, sum(pattern), ticks_per_launch=9)  # launches of 9, 9, 6 ticks: same trajectory, other chunking
    assert added == B * sum(pattern) == len(rows)
    assert buf.size == max_size and buf.idx == len(rows) % max_size
    st, nx, ac, rw, dd, im = (np_(x) for x in (buf.states, buf.next_states, buf.actions, buf.rewards, buf.dones, buf.imposters))
    for k in range(len(rows) - max_size, len(rows)):
        p = k % max_size
        w, a, r, nw, d, i = rows[k]
        assert np.array_equal(st[p], w) and np.array_equal(nx[p], nw), f"row {k} windows"
        assert np.array_equal(ac[p], a) and np.array_equal(rw[p], r) and bool(dd[p, 0]) == d and np.array_equal(im[p], i), f"row {k}"


def test_device_replay_populate(pkg):
    """Batched ReplayBuffer.populate (src/replay_memory.py:96-143): window roll, ring layout, episode boundaries."""
    B, T = 256, 3
    env = pkg.BatchedFourRoomEnv(1, 2, 2, batch=B, auto_reset=True, seed=8, obs=pkg.ObsConfig("raw"), max_time_steps=30)
    S = env.flattened_state_size
    buf = pkg.DeviceReplayBuffer(B * 50, S, T, env.n_agents, env.n_imposters, device=env.device)
    added = buf.populate(env, 40)
    assert added == B * 40 and buf.size == B * 40 and buf.idx == B * 40
    n = B * 40
    st = buf.states[:n].view(40, B, T, S)
    nx = buf.next_states[:n].view(40, B, T, S)
    dn = buf.dones[:n].view(40, B)
    # the window rolls: next_states[:, :-1] == states[:, 1:] for every transition
    assert torch.equal(nx[:, :, :-1], st[:, :, 1:])
    # the next tick starts from the rolled window unless the episode ended (then it is T copies of the new first state)
    rew_done = dn[:-1]
    cont = ~rew_done
    same = (st[1:] == nx[:-1]).flatten(2).all(-1)
    assert bool(same[cont].float().mean() > 0.95)  # truncated episodes also restart; done-free rows continue exactly
    fresh = st[1:][rew_done]
    assert bool((fresh == fresh[:, :1]).all())
    assert buf.actions[:n].max() < 7 and buf.imposters[:n].min() >= 0 and buf.imposters[:n].max() < env.n_agents


def test_device_replay_populate_exact_terminal_states(pkg):
    """With auto_reset=False populate() keeps the reference's terminal semantics (replay_memory.py:120-136): the
    next_states window of a transition that ended its episode ends with the TRUE terminal state (here: 1v1, `done`
    means the crew member was killed, so its alive flag is 0 there), and the following transition of that env starts
    from T copies of a fresh state (everybody alive)."""
    B, T, steps = 512, 2, 120
    env = pkg.BatchedImposterTrainingGround(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0,
                                            time_step_reward=0, include_walls=False, batch=B, auto_reset=False, seed=4,
                                            obs=pkg.ObsConfig("raw"), max_time_steps=40)
    S = env.flattened_state_size
    buf = pkg.DeviceReplayBuffer(B * steps, S, T, 2, 1, device=env.device)
    assert buf.populate(env, steps) == B * steps
    st = buf.states.view(steps, B, T, S)
    nx = buf.next_states.view(steps, B, T, S)
    dn = buf.dones.view(steps, B)
    assert int(dn.sum()) > 50
    crew_alive_after = nx[:, :, -1, 5]  # flatten_state: x0 y0 x1 y1 alive0 alive1
    assert bool((crew_alive_after[dn] == 0).all()) and bool((crew_alive_after[~dn] == 1).all())
    after_end = st[1:][dn[:-1]]
    assert bool((after_end[:, :, 4:6] == 1).all()) and bool((after_end == after_end[:, :1]).all())
    # truncation (t reaches max_time_steps - 1 = 39): episode restarts although done is False
    fresh = (st[1:] == st[1:, :, :1]).flatten(2).all(-1) & (st[1:] != nx[:-1]).flatten(2).any(-1)
    assert int((fresh & ~dn[:-1]).sum()) > 0
