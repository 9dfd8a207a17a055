#!/usr/bin/env python3
"""Generate golden traces by running the UNMODIFIED reference env (container only).

    python tests/golden/generate_golden.py            # rewrites tests/golden/*.npz

The reference (`/root/reference/src/environment/{base,pred_prey,tagging}.py`) is imported through the
in-process shims in `_refshim.py` and driven with the `ReplayBuffer.populate`-shaped loop
(replay_memory.py:96-143): `reset(seed)`; repeat {`sample_actions()`; `step()`; `reset()` on
done|truncated}.  Only DATA is written: inputs (ctor kwargs, grid, seed, actions, injected states) and
the reference's outputs (state, rewards, done, truncated, the 13 `info` counters) per step, plus the
number of raw MT19937 words numpy consumed so far (so a restatement's stream position can be checked).

Every fixture is one `.npz`:
  meta            json string: class, kwargs, grid, seed, mode, notes
  ep_start[S]     1 if `reset()` was called right before this step
  words[S+1]      cumulative raw 32-bit words drawn from the global MT19937 after reset / each step
                  (words[0] = after the first reset); sampled mode includes sample_actions draws
  actions[S,A]    role-relative action indices passed to step()
  pre_pos[S,A,2], pre_alive[S,A], pre_jobpos[S,J,2], pre_jobdone[S,J], pre_imp[S,A] (imposter mask),
  pre_t[S]        state BEFORE the step (after any reset / injection)
  pos[S,A,2], alive[S,A], jobdone[S,J]           state AFTER the step
  rewards[S,A] float64 (bit pattern preserved, -0.0 included), done[S], trunc[S], metrics[S,13] int32
  order[S,A]      agent order used by the step (identity when not shuffled)
  tagging only:   pre_used/pre_counts/pre_timer and used[S,A], counts[S,A], timer_left[S]
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()

from src.environment import FourRoomEnv, FourRoomEnvWithTagging, ImposterTrainingGround  # noqa: E402
from src.metrics import SusMetrics  # noqa: E402

METRIC_ORDER = [m.value for m in SusMetrics]


# ----------------------------------------------------------------------------------------------
# grids the reference does not define (BASELINE.json configs 3/4 ask for 14x14) -- build-defined
# ----------------------------------------------------------------------------------------------
def four_room_grid(n: int, wall: int, doors) -> np.ndarray:
    """Transpose-symmetric four-room layout: wall row+column at index `wall`, door gaps at `doors`."""
    g = np.ones((n, n), dtype=bool)
    for i in range(n):
        if i not in doors:
            g[i, wall] = 0
            g[wall, i] = 0
    return g


GRID14 = four_room_grid(14, 6, (2, 10))


def asym_grid(n: int, seed: int) -> np.ndarray:
    """Deliberately NOT transpose-symmetric: exposes grid[y,x] (move) vs grid[x,y] (spawn)."""
    rs = np.random.RandomState(seed)
    g = rs.rand(n, n) > 0.25
    g[0, 0] = True
    return g


def words_drawn_since(state0, state1) -> int:
    """Raw MT19937 words consumed between two np.random.get_state() snapshots (< 624 apart, or
    multiples handled by key comparison)."""
    k0, p0 = state0[1], state0[2]
    k1, p1 = state1[1], state1[2]
    if np.array_equal(k0, k1):
        return p1 - p0
    # one or more twists happened; step a copy forward to count
    rs = np.random.RandomState()
    rs.set_state(state0)
    n = 0
    while True:
        st = rs.get_state()
        if st[2] == p1 and np.array_equal(st[1], k1):
            return n
        rs.randint(0, 2**32, dtype=np.uint32)  # consumes exactly one word
        n += 1
        if n > 200000:
            raise RuntimeError("could not sync MT19937 state")


class Recorder:
    def __init__(self, env, tagging: bool):
        self.env = env
        self.tagging = tagging
        self.rows = {k: [] for k in (
            "ep_start actions pre_pos pre_alive pre_jobpos pre_jobdone pre_imp pos alive jobdone "
            "rewards done trunc metrics order pre_t").split()}
        if tagging:
            for k in "pre_used pre_counts pre_timer used counts timer_left".split():
                self.rows[k] = []
        self.words = []
        self._last_state = None
        self._total = 0

    def mark(self):
        st = np.random.get_state()
        if self._last_state is not None:
            self._total += words_drawn_since(self._last_state, st)
        self._last_state = st
        return self._total

    def start(self):
        self._last_state = np.random.get_state()
        self._total = 0

    def step(self, actions, ep_start: bool):
        env = self.env
        r = self.rows
        r["ep_start"].append(int(ep_start))
        r["actions"].append(np.array(actions, dtype=np.int16))
        r["pre_pos"].append(env.agent_positions.copy())
        r["pre_alive"].append(env.alive_agents.copy())
        r["pre_jobpos"].append(env.job_positions.copy().reshape(-1, 2))
        r["pre_jobdone"].append(env.completed_jobs.copy())
        r["pre_imp"].append(env.imposter_mask.copy())
        r["pre_t"].append(int(env.t))
        if self.tagging:
            r["pre_used"].append(env.used_tag_actions.copy())
            r["pre_counts"].append(env.tag_counts.copy())
            r["pre_timer"].append(env.tag_reset_timer)
        # capture the order the env used
        order_seen = []
        orig_shuffle = np.random.shuffle

        def rec_shuffle(x):
            orig_shuffle(x)
            order_seen.append(list(x))

        np.random.shuffle = rec_shuffle
        try:
            out = env.step(actions)
        finally:
            np.random.shuffle = orig_shuffle
        state, rewards, done, trunc, info = out
        r["order"].append(np.array(order_seen[0] if order_seen else list(range(env.n_agents)), dtype=np.int8))
        r["pos"].append(env.agent_positions.copy())
        r["alive"].append(env.alive_agents.copy())
        r["jobdone"].append(env.completed_jobs.copy())
        r["rewards"].append(np.array(rewards, dtype=np.float64))
        r["done"].append(int(done))
        r["trunc"].append(int(trunc))
        r["metrics"].append(np.array([info[m] for m in SusMetrics], dtype=np.int32))
        if self.tagging:
            r["used"].append(env.used_tag_actions.copy())
            r["counts"].append(env.tag_counts.copy())
            r["timer_left"].append(state[6])
        self.words.append(self.mark())
        return done, trunc

    def arrays(self):
        out = {}
        for k, v in self.rows.items():
            a = np.array(v)
            if a.dtype == bool:
                a = a.astype(np.uint8)
            elif a.dtype == np.int64:
                a = a.astype(np.int16)
            out[k] = a
        out["words"] = np.array(self.words, dtype=np.int64)
        return out


def make_env(spec):
    cls = {"base": FourRoomEnv, "itg": ImposterTrainingGround, "tagging": FourRoomEnvWithTagging}[spec["class"]]
    env = cls(**spec["kwargs"])
    grid = spec.get("grid")
    if grid is not None:  # instance override for NxN grids the reference hard-codes as 9x9
        g = np.array(grid, dtype=bool)
        env.grid = g
        env.valid_positions = np.argwhere(g)
        env.n_rows = env.n_cols = g.shape[0]
    return env


def run_sampled(spec, seed, steps, continue_after_end=0):
    """populate()-shaped loop; `continue_after_end` extra steps are taken WITHOUT reset after the first
    episode end (the reference lets callers keep stepping: t saturates, TOTAL_TIME_STEPS keeps counting)."""
    env = make_env(spec)
    rec = Recorder(env, spec["class"] == "tagging")
    np.random.seed(seed)
    rec.start()
    env.reset()
    words0 = rec.mark()
    ep_start = True
    coasting = continue_after_end
    for _ in range(steps):
        a = env.sample_actions()
        done, trunc = rec.step(a, ep_start)
        ep_start = False
        if done or trunc:
            if coasting > 0:
                coasting -= 1
                continue
            env.reset()
            rec.words[-1] = rec.mark()
            ep_start = True
    arrs = rec.arrays()
    arrs["words"] = np.concatenate([[words0], arrs["words"]])
    return env, arrs


def run_given(spec, seed, script):
    """Directed trace. `script` = list of dicts {inject: {...}|None, actions: [...]}; global RNG seeded once."""
    env = make_env(spec)
    tagging = spec["class"] == "tagging"
    rec = Recorder(env, tagging)
    np.random.seed(seed)
    rec.start()
    env.reset()
    words0 = rec.mark()
    first = True
    for item in script:
        inj = item.get("inject")
        if inj:
            if "pos" in inj:
                env.agent_positions[...] = np.array(inj["pos"])
            if "alive" in inj:
                env.alive_agents[...] = np.array(inj["alive"], dtype=bool)
            if "jobpos" in inj:
                env.job_positions[...] = np.array(inj["jobpos"])
            if "jobdone" in inj:
                env.completed_jobs[...] = np.array(inj["jobdone"], dtype=bool)
            if "t" in inj:
                env.t = int(inj["t"])
            if tagging:
                if "used" in inj:
                    env.used_tag_actions[...] = np.array(inj["used"], dtype=bool)
                if "counts" in inj:
                    env.tag_counts[...] = np.array(inj["counts"])
                if "timer" in inj:
                    env.tag_reset_timer = int(inj["timer"])
        rec.step(np.array(item["actions"]), first)
        first = False
    arrs = rec.arrays()
    arrs["words"] = np.concatenate([[words0], arrs["words"]])
    arrs["inject"] = np.array([1 if it.get("inject") else 0 for it in script], dtype=np.uint8)
    return env, arrs


def save(name, spec, seed, mode, env, arrs, notes=""):
    meta = dict(spec)
    meta.update(seed=int(seed), mode=mode, notes=notes, metric_order=METRIC_ORDER,
                n_agents=int(env.n_agents), n_imposters=int(env.n_imposters), n_jobs=int(env.n_jobs),
                grid_used=np.asarray(env.grid, dtype=np.uint8).tolist(),
                imposter_idxs_final=[int(i) for i in env.imposter_idxs])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrs)
    print(f"{name}: S={len(arrs['done'])} episodes={int(arrs['ep_start'].sum())} "
          f"kills={int(arrs['metrics'][:,0].max())} size={os.path.getsize(path)} B")


ITG_NB = dict(n_crew=1, n_jobs=0, kill_reward=-3, sabotage_reward=0, end_of_game_reward=0, time_step_reward=0)


def main():
    # ---------------- sampled (populate-shaped) traces --------------------------------------
    sampled = [
        # name, spec, seeds, steps
        ("itg_1v1_nowalls", dict(**{"class": "itg"}, kwargs=dict(ITG_NB, include_walls=False)), range(4), 700),
        ("itg_1v1_walls", dict(**{"class": "itg"}, kwargs=dict(ITG_NB)), range(2), 700),
        ("itg_1v3_j2", dict(**{"class": "itg"}, kwargs=dict(n_crew=3, n_jobs=2, kill_reward=-3, sabotage_reward=1,
                                                          end_of_game_reward=7, time_step_reward=-1)), range(3), 500),
        ("itg_1v2_shuffleimp", dict(**{"class": "itg"}, kwargs=dict(n_crew=2, n_jobs=1, kill_reward=-3, sabotage_reward=1,
                                                                 end_of_game_reward=5, time_step_reward=0,
                                                                 shuffle_imposter_index=True)), range(2), 400),
        ("base_1v2_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4)), range(4), 600),
        ("base_2v6_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=6, n_jobs=4)), range(4), 600),
        ("base_1v3_j5_fixedorder", dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=3, n_jobs=5,
                                                                       is_action_order_random=False,
                                                                       shuffle_imposter_index=False)), range(2), 500),
        ("base_2v3_j3_tsr", dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=3, n_jobs=3, time_step_reward=-1,
                                                                kill_reward=-4, dead_penalty=-7, game_end_reward=20,
                                                                complete_job_reward=2, sabotage_reward=6)), range(2), 500),
        ("base_1v2_j0_instantwin", dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=0)), range(1), 40),
        ("base_1v2_j4_nowalls_short", dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4,
                                                                          include_walls=False, max_time_steps=25)), range(2), 300),
        ("base14_1v2_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4),
                               grid=GRID14.astype(int).tolist()), range(4), 600),
        ("base14_2v6_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=6, n_jobs=4),
                               grid=GRID14.astype(int).tolist()), range(4), 600),
        ("base14_asym_1v2_j4", dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=4),
                                    grid=asym_grid(14, 7).astype(int).tolist()), range(2), 500),
        ("base11_asym_2v4_j3", dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=4, n_jobs=3),
                                    grid=asym_grid(11, 3).astype(int).tolist()), range(2), 400),
        ("tagging_1v4_j5", dict(**{"class": "tagging"}, kwargs=dict(n_imposters=1, n_crew=4, n_jobs=5)), range(3), 600),
        ("tagging_2v5_j3_int7", dict(**{"class": "tagging"}, kwargs=dict(n_imposters=2, n_crew=5, n_jobs=3,
                                                                        tag_reset_interval=7, vote_reward=4)), range(3), 600),
        ("tagging_1v2_j2_int3_tsr", dict(**{"class": "tagging"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=2,
                                                                            tag_reset_interval=3, vote_reward=3,
                                                                            time_step_reward=-1, dead_penalty=-2)), range(2), 400),
        ("tagging14_2v6_j4", dict(**{"class": "tagging"}, kwargs=dict(n_imposters=2, n_crew=6, n_jobs=4, tag_reset_interval=20),
                                  grid=GRID14.astype(int).tolist()), range(2), 500),
    ]
    for name, spec, seeds, steps in sampled:
        for seed in seeds:
            env, arrs = run_sampled(spec, seed, steps)
            save(f"{name}_s{seed}", spec, seed, "sampled", env, arrs)

    # keep stepping after the episode ended (no reset): t saturates at max-1, counters keep counting
    spec = dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=3, max_time_steps=12))
    env, arrs = run_sampled(spec, 5, 60, continue_after_end=30)
    save("base_1v2_j3_coast_s5", spec, 5, "sampled", env, arrs, notes="30 steps taken after first episode end without reset")
    spec = dict(**{"class": "tagging"}, kwargs=dict(n_imposters=1, n_crew=3, n_jobs=2, max_time_steps=15, tag_reset_interval=4))
    env, arrs = run_sampled(spec, 6, 80, continue_after_end=25)
    save("tagging_1v3_j2_coast_s6", spec, 6, "sampled", env, arrs, notes="25 steps after first episode end without reset")

    # ---------------- directed traces ---------------------------------------------------------
    # base 1v3 (imposter index 0 forced), everyone on one cell: multi-candidate kill draws, fix-then-killed,
    # sabotage toggling, _merge_rewards' [:n_imposters] negation with a shuffled imposter index.
    K_IMP_KILL, K_IMP_SAB, K_CREW_FIX = 6, 5, 5
    spec = dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=3, n_jobs=2, shuffle_imposter_index=False,
                                                is_action_order_random=True))
    script = []
    for rep in range(24):
        script.append(dict(inject=dict(pos=[[1, 1]] * 4, alive=[1, 1, 1, 1], jobpos=[[1, 1], [7, 7]],
                                       jobdone=[rep % 2, 0]),
                           actions=[K_IMP_KILL if rep % 3 else K_IMP_SAB, K_CREW_FIX, rep % 5, K_CREW_FIX]))
        script.append(dict(inject=None, actions=[K_IMP_KILL, K_CREW_FIX, 0, 0]))
        script.append(dict(inject=None, actions=[K_IMP_KILL, 0, 0, 0]))
    env, arrs = run_given(spec, 11, script)
    save("directed_base_1v3_pile", spec, 11, "given", env, arrs, notes="multi-candidate kills, fix-then-killed, sabotage toggle")

    spec = dict(**{"class": "base"}, kwargs=dict(n_imposters=2, n_crew=5, n_jobs=3, shuffle_imposter_index=True))
    script = []
    rs = np.random.RandomState(99)
    for rep in range(40):
        script.append(dict(inject=dict(pos=[[3, 3]] * 7, alive=[1] * 7, jobpos=[[3, 3], [0, 0], [8, 8]],
                                       jobdone=[int(rs.randint(2)), 0, 1]),
                           actions=[int(rs.randint(5, 7)) for _ in range(7)]))
        script.append(dict(inject=None, actions=[int(rs.randint(0, 6)) for _ in range(7)]))
    # NOTE: role-relative index 6 is only valid for imposters; indices 0..5 are valid for both roles.
    # First item of each pair uses 5 or 6 for everyone: 6 raises IndexError for crew in the reference, so
    # clamp crew to 5 (FIX) using the env's own role assignment after reset.
    env = make_env(spec)
    np.random.seed(12)
    env.reset()
    imp = set(int(i) for i in env.imposter_idxs)
    for it in script:
        it["actions"] = [a if (i in imp or a < 6) else 5 for i, a in enumerate(it["actions"])]
    env, arrs = run_given(spec, 12, script)
    save("directed_base_2v5_pile", spec, 12, "given", env, arrs, notes="two imposters killing in one step, shuffled roles")

    # ITG 1v4 pile: fixed order, multi-candidate kill
    spec = dict(**{"class": "itg"}, kwargs=dict(n_crew=4, n_jobs=0, kill_reward=-3, sabotage_reward=0,
                                               end_of_game_reward=9, time_step_reward=0))
    script = []
    for rep in range(20):
        script.append(dict(inject=dict(pos=[[2, 2]] * 5, alive=[1, 1, 1, rep % 2, 1]), actions=[5, 0, 0, 0, 0]))
        for _ in range(4):
            script.append(dict(inject=None, actions=[5, 0, 0, 0, 0]))
    env, arrs = run_given(spec, 13, script)
    save("directed_itg_1v4_pile", spec, 13, "given", env, arrs, notes="ITG multi-candidate kill until imposter wins, then keeps stepping")

    # truncation boundary
    spec = dict(**{"class": "base"}, kwargs=dict(n_imposters=1, n_crew=2, n_jobs=2, max_time_steps=1000))
    script = [dict(inject=dict(t=996) if i == 0 else None, actions=[0, 0, 0]) for i in range(8)]
    env, arrs = run_given(spec, 14, script)
    save("directed_base_trunc999", spec, 14, "given", env, arrs, notes="t injected to 996; truncated at t==999 and every step after")

    # tagging: tie -> lowest index, quorum even/odd alive, dead voter, tag on dead target, imposter voted out
    spec = dict(**{"class": "tagging"}, kwargs=dict(n_imposters=1, n_crew=4, n_jobs=2, tag_reset_interval=2, vote_reward=3,
                                                   shuffle_imposter_index=False, is_action_order_random=True))
    # role-relative: imposter (agent 0) 7 role actions then others [1,2,3,4]; crew i: 6 role actions then others ascending
    def tag(agent, target, imp=0):
        others = [j for j in range(5) if j != agent]
        base = 7 if agent == imp else 6
        return base + others.index(target)
    script = [
        dict(inject=dict(pos=[[0, 0], [8, 8], [0, 8], [8, 0], [5, 5]], alive=[1, 1, 1, 1, 1], timer=0, counts=[0] * 5, used=[0] * 5),
             actions=[tag(0, 1), tag(1, 2), tag(2, 1), tag(3, 2), 0]),          # 1 and 2 tie at 2 votes
        dict(inject=None, actions=[0, 0, 0, 0, tag(4, 2)]),                         # vote fires: 2 has 3 >= quorum 3
        dict(inject=None, actions=[tag(0, 2), tag(1, 0), tag(3, 0), tag(3, 0), tag(4, 0)]),  # tag dead target fails; imposter tagged
        dict(inject=None, actions=[0, 0, tag(2, 0), 0, 0]),                         # dead voter 2 still votes -> imposter out
        dict(inject=dict(alive=[1, 1, 0, 1, 1], timer=0, counts=[0] * 5, used=[0] * 5), actions=[tag(0, 1), tag(1, 3), 0, tag(3, 1), 0]),
        dict(inject=None, actions=[0, 0, 0, 0, 0]),                                 # 4 alive: quorum 2; agent1 has 2 -> crew voted out
        dict(inject=dict(alive=[1, 1, 1, 1, 1], timer=0, counts=[0] * 5, used=[0] * 5), actions=[tag(0, 3), 0, 0, 0, tag(4, 1)]),
        dict(inject=None, actions=[0, 0, 0, 0, 0]),                                 # tie 1 vs 3 with 1 vote each < quorum 3: nobody out
    ] * 3
    env, arrs = run_given(spec, 15, script)
    save("directed_tagging_votes", spec, 15, "given", env, arrs, notes="ties, quorum parity, dead voter, tag on dead, imposter voted out")


if __name__ == "__main__":
    main()
