"""ctypes front-end of the CPU oracle (oracle/susnet_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (sus-net_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libsusnet_oracle.so")

MAX_AGENTS, MAX_JOBS, MAX_GRID, N_METRICS = 16, 16, 16, 13
VARIANTS = {"base": 0, "itg": 1, "tagging": 2}
METRIC_NAMES = [
    "imp_killed_crew", "imp_voted_out", "crew_voted_out", "sabotaged_jobs", "completed_jobs",
    "total_stalemates", "total_time_steps", "imposter_won", "crew_won", "avg_crew_returns",
    "avg_imposter_returns", "crew_loss", "imposter_loss",
]
FLAT = {"onehot_pos": 0, "coord_pos": 1, "alive_crew": 2, "l1_crew": 3, "closest_crew": 4, "walls3x3": 5,
        "dist_to_imp": 6, "room_loc": 7, "scent": 8}


class SoConfig(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("n_imposters", C.c_int32), ("n_crew", C.c_int32), ("n_jobs", C.c_int32),
        ("grid_n", C.c_int32), ("grid", (C.c_uint8 * MAX_GRID) * MAX_GRID),
        ("kill_reward", C.c_double), ("complete_job_reward", C.c_double), ("sabotage_reward", C.c_double),
        ("time_step_reward", C.c_double), ("game_end_reward", C.c_double), ("dead_penalty", C.c_double),
        ("vote_reward", C.c_double), ("max_time_steps", C.c_int32), ("is_action_order_random", C.c_int32),
        ("shuffle_imposter_index", C.c_int32), ("tag_reset_interval", C.c_int32),
    ]


class SoRng(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("mt", C.c_uint32 * 624), ("mti", C.c_int32), ("tape", C.POINTER(C.c_uint32)),
        ("tape_len", C.c_int64), ("seed", C.c_uint64), ("env_id", C.c_uint64), ("cursor", C.c_uint64),
        ("tick", C.c_uint64), ("overflow", C.c_int32), ("episode", C.c_uint32), ("reset_pos", C.c_uint32), ("in_reset", C.c_int32),
    ]


class SoEnv(C.Structure):
    _fields_ = [
        ("cfg", SoConfig), ("rng", SoRng), ("A", C.c_int32), ("J", C.c_int32), ("n_valid", C.c_int32),
        ("valid", (C.c_uint8 * 2) * (MAX_GRID * MAX_GRID)),
        ("pos", (C.c_int32 * 2) * MAX_AGENTS), ("alive", C.c_int32 * MAX_AGENTS),
        ("imp_mask", C.c_int32 * MAX_AGENTS), ("imp_idxs", C.c_int32 * MAX_AGENTS),
        ("jobpos", (C.c_int32 * 2) * MAX_JOBS), ("jobdone", C.c_int32 * MAX_JOBS),
        ("used", C.c_int32 * MAX_AGENTS), ("counts", C.c_int32 * MAX_AGENTS), ("timer", C.c_int32),
        ("t", C.c_int32), ("n_role_actions", C.c_int32 * MAX_AGENTS), ("metrics", C.c_int64 * N_METRICS),
        ("rewards", C.c_double * MAX_AGENTS), ("order", C.c_int32 * MAX_AGENTS),
        ("aw_W", C.c_int32), ("aw_tpw", C.c_int32), ("aw_word", C.c_uint8 * (2 * MAX_AGENTS)),
    ]


_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few hundred ms). Safe to call repeatedly."""
    src = os.path.join(HERE, "susnet_oracle.c")
    hdr = os.path.join(HERE, "susnet_oracle.h")
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return LIB_PATH
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        P = C.POINTER
        L.so_env_init.argtypes = [P(SoEnv), P(SoConfig)]
        L.so_env_init.restype = C.c_int
        L.so_seed_mt.argtypes = [P(SoEnv), C.c_uint32]
        L.so_set_tape.argtypes = [P(SoEnv), P(C.c_uint32), C.c_int64]
        L.so_set_philox.argtypes = [P(SoEnv), C.c_uint64, C.c_uint64, C.c_uint64]
        L.so_set_tick.argtypes = [P(SoEnv), C.c_uint64]
        L.so_philox4x32_10.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
        L.so_next_u32.argtypes = [P(SoEnv)]
        L.so_next_u32.restype = C.c_uint32
        L.so_reset.argtypes = [P(SoEnv)]
        L.so_sample_actions.argtypes = [P(SoEnv), P(C.c_int32)]
        L.so_step.argtypes = [P(SoEnv), P(C.c_int32), P(C.c_double), P(C.c_int32), P(C.c_int32)]
        L.so_step.restype = C.c_int
        L.so_n_actions.argtypes = [P(SoEnv), C.c_int]
        L.so_n_actions.restype = C.c_int
        L.so_sizeof_env.restype = C.c_int
        L.so_batch_reset.argtypes = [P(SoEnv), C.c_int64, C.c_int]
        L.so_batch_step.argtypes = [P(SoEnv), C.c_int64, P(C.c_int32), P(C.c_double), P(C.c_uint8), P(C.c_uint8), C.c_int]
        L.so_batch_step.restype = C.c_int
        L.so_batch_random_rollout.argtypes = [P(SoEnv), C.c_int64, C.c_int64, C.c_int, P(C.c_int64), P(C.c_double)]
        L.so_batch_random_rollout.restype = C.c_int64
        L.so_batch_sample_actions.argtypes = [P(SoEnv), C.c_int64, C.c_void_p]
        L.so_batch_reset_masked.argtypes = [P(SoEnv), C.c_int64, C.c_void_p]
        L.so_batch_obs_raw.argtypes = [P(SoEnv), C.c_int64, C.c_void_p]
        L.so_batch_export.argtypes = [P(SoEnv), C.c_int64] + [C.c_void_p] * 11
        L.so_batch_export.restype = None
        L.so_obs_flat_size.argtypes = [P(SoEnv), P(C.c_int32), C.c_int]
        L.so_obs_flat_size.restype = C.c_int
        L.so_obs_flat.argtypes = [P(SoEnv), P(C.c_int32), C.c_int, P(C.c_float)]
        L.so_obs_flat.restype = C.c_int
        L.so_obs_planes.argtypes = [P(SoEnv), P(C.c_float), P(C.c_float)]
        L.so_obs_raw.argtypes = [P(SoEnv), P(C.c_double)]
        L.so_obs_raw_size.argtypes = [P(SoEnv)]
        L.so_obs_raw_size.restype = C.c_int
        assert L.so_sizeof_env() == C.sizeof(SoEnv), (L.so_sizeof_env(), C.sizeof(SoEnv))
        _lib = L
    return _lib


REFERENCE_WALLS_9 = [(0, 4), (2, 4), (3, 4), (4, 4), (5, 4), (6, 4), (8, 4), (4, 0), (4, 2), (4, 3), (4, 5), (4, 6), (4, 8)]


def reference_grid(include_walls: bool = True) -> np.ndarray:
    """The reference's hard-coded 9x9 four-room grid (src/environment/base.py:171-197), as data."""
    g = np.ones((9, 9), dtype=np.uint8)
    if include_walls:
        for i, j in REFERENCE_WALLS_9:
            g[i, j] = 0
    return g


def make_config(variant: str, *, n_imposters=1, n_crew=1, n_jobs=0, grid=None, include_walls=True,
                kill_reward=-5, complete_job_reward=3, sabotage_reward=3, time_step_reward=0,
                game_end_reward=10, dead_penalty=-2, vote_reward=3, max_time_steps=1000,
                is_action_order_random=True, shuffle_imposter_index=True, tag_reset_interval=50,
                end_of_game_reward=None) -> SoConfig:
    """Mirror of the reference ctor kwargs (base.py:103-120, pred_prey.py:26-38, tagging.py:10-12)."""
    cfg = SoConfig()
    cfg.variant = VARIANTS[variant]
    if variant == "itg":
        n_imposters = 1
        if end_of_game_reward is not None:
            game_end_reward = end_of_game_reward
    cfg.n_imposters, cfg.n_crew, cfg.n_jobs = n_imposters, n_crew, n_jobs
    g = reference_grid(include_walls) if grid is None else np.asarray(grid, dtype=np.uint8)
    n = g.shape[0]
    assert g.shape == (n, n) and n <= MAX_GRID
    cfg.grid_n = n
    for i in range(n):
        for j in range(n):
            cfg.grid[i][j] = int(g[i, j] != 0)
    cfg.kill_reward, cfg.complete_job_reward, cfg.sabotage_reward = kill_reward, complete_job_reward, sabotage_reward
    cfg.time_step_reward, cfg.game_end_reward, cfg.dead_penalty = time_step_reward, game_end_reward, dead_penalty
    cfg.vote_reward = vote_reward
    cfg.max_time_steps = max_time_steps
    cfg.is_action_order_random = int(is_action_order_random)
    cfg.shuffle_imposter_index = int(shuffle_imposter_index)
    cfg.tag_reset_interval = tag_reset_interval
    return cfg


def config_from_fixture_meta(meta: dict) -> SoConfig:
    """Build the oracle config from a golden fixture's recorded reference ctor kwargs."""
    kw = dict(meta["kwargs"])
    cls = meta["class"]
    grid = np.array(meta["grid_used"], dtype=np.uint8)
    kw.pop("include_walls", None)
    kw.pop("debug", None)
    if cls == "itg":
        kw.setdefault("shuffle_imposter_index", False)  # pred_prey.py:36
    return make_config(cls, grid=grid, **kw)


class OracleBatch:
    """B independent oracle envs (one C struct each)."""

    def __init__(self, cfg: SoConfig, batch: int):
        self.L = lib()
        self.B = batch
        self.envs = (SoEnv * batch)()
        for b in range(batch):
            rc = self.L.so_env_init(C.byref(self.envs[b]), C.byref(cfg))
            if rc != 0:
                raise ValueError(f"oracle rejected config (rc={rc})")
        self.A = self.envs[0].A
        self.J = self.envs[0].J
        self.N = self.envs[0].cfg.grid_n
        self.variant = self.envs[0].cfg.variant
        self._tapes = None

    # -- word sources -------------------------------------------------------------------------
    def seed_mt(self, seeds):
        for b, s in enumerate(seeds):
            self.L.so_seed_mt(C.byref(self.envs[b]), int(s) & 0xFFFFFFFF)

    def set_tapes(self, tapes: np.ndarray):
        tapes = np.ascontiguousarray(tapes, dtype=np.uint32)
        assert tapes.shape[0] == self.B
        self._tapes = tapes
        for b in range(self.B):
            p = tapes[b].ctypes.data_as(C.POINTER(C.c_uint32))
            self.L.so_set_tape(C.byref(self.envs[b]), p, tapes.shape[1])

    def set_philox(self, seed: int, env_id_base: int = 0, cursor: int = 0):
        for b in range(self.B):
            self.L.so_set_philox(C.byref(self.envs[b]), seed, env_id_base + b, cursor)

    def set_tick(self, tick: int):
        for b in range(self.B):
            self.L.so_set_tick(C.byref(self.envs[b]), tick)

    def raw_words(self, n: int) -> np.ndarray:
        """Draw n raw words from every env's source (advances it)."""
        out = np.empty((self.B, n), dtype=np.uint32)
        for b in range(self.B):
            for k in range(n):
                out[b, k] = self.L.so_next_u32(C.byref(self.envs[b]))
        return out

    # -- dynamics -------------------------------------------------------------------------------
    def reset(self, mask=None, threads=1):
        if mask is None:
            self.L.so_batch_reset(self.envs, self.B, threads)
        else:
            m = np.ascontiguousarray(np.asarray(mask), dtype=np.uint8)
            self.L.so_batch_reset_masked(self.envs, self.B, m.ctypes.data)

    def sample_actions(self) -> np.ndarray:
        out = np.zeros((self.B, self.A), dtype=np.int32)
        self.L.so_batch_sample_actions(self.envs, self.B, out.ctypes.data)
        return out

    def step(self, actions, threads=1):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.B, self.A)
        rew = np.zeros((self.B, self.A), dtype=np.float64)
        done = np.zeros(self.B, dtype=np.uint8)
        trunc = np.zeros(self.B, dtype=np.uint8)
        rc = self.L.so_batch_step(self.envs, self.B, a.ctypes.data_as(C.POINTER(C.c_int32)),
                                  rew.ctypes.data_as(C.POINTER(C.c_double)),
                                  done.ctypes.data_as(C.POINTER(C.c_uint8)),
                                  trunc.ctypes.data_as(C.POINTER(C.c_uint8)), threads)
        return rew, done, trunc, rc

    def step_one(self, b: int, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        rew = np.zeros(self.A, dtype=np.float64)
        d, t = C.c_int32(0), C.c_int32(0)
        rc = self.L.so_step(C.byref(self.envs[b]), a.ctypes.data_as(C.POINTER(C.c_int32)),
                            rew.ctypes.data_as(C.POINTER(C.c_double)), C.byref(d), C.byref(t))
        return rew, bool(d.value), bool(t.value), rc

    def random_rollout(self, steps: int, threads: int = 1):
        ep, rs = C.c_int64(0), C.c_double(0)
        n = self.L.so_batch_random_rollout(self.envs, self.B, steps, threads, C.byref(ep), C.byref(rs))
        return int(n), int(ep.value), float(rs.value)

    # -- state access ---------------------------------------------------------------------------
    def _field(self, name, width=None, dtype=np.int64):
        n = width
        out = np.zeros((self.B, n) if n is not None else (self.B,), dtype=dtype)
        for b in range(self.B):
            v = getattr(self.envs[b], name)
            out[b] = np.ctypeslib.as_array(v)[:n] if n is not None else v
        return out

    @property
    def pos(self):
        out = np.zeros((self.B, self.A, 2), dtype=np.int64)
        for b in range(self.B):
            out[b] = np.ctypeslib.as_array(self.envs[b].pos)[: self.A]
        return out

    @property
    def jobpos(self):
        out = np.zeros((self.B, self.J, 2), dtype=np.int64)
        for b in range(self.B):
            out[b] = np.ctypeslib.as_array(self.envs[b].jobpos)[: self.J]
        return out

    alive = property(lambda s: s._field("alive", s.A))
    imp_mask = property(lambda s: s._field("imp_mask", s.A))
    jobdone = property(lambda s: s._field("jobdone", s.J))
    used = property(lambda s: s._field("used", s.A))
    counts = property(lambda s: s._field("counts", s.A))
    timer = property(lambda s: s._field("timer"))
    t = property(lambda s: s._field("t"))
    metrics = property(lambda s: s._field("metrics", N_METRICS))
    order = property(lambda s: s._field("order", s.A))

    @property
    def cursor(self):
        return np.array([self.envs[b].rng.cursor for b in range(self.B)], dtype=np.int64)

    @property
    def episode(self):
        """Resets drawn so far per env (index of the production RESET stream; Philox kind)."""
        return np.array([self.envs[b].rng.episode for b in range(self.B)], dtype=np.int64)

    def set_episode(self, episode):
        ep = np.broadcast_to(np.asarray(episode, dtype=np.int64), (self.B,))
        for b in range(self.B):
            self.envs[b].rng.episode = int(ep[b])

    @property
    def tape_overflow(self):
        return np.array([self.envs[b].rng.overflow for b in range(self.B)], dtype=np.int64)

    def export(self) -> dict:
        """Dense numpy snapshot of every env (one C call)."""
        B, A, J = self.B, self.A, self.J
        out = dict(pos=np.zeros((B, A, 2), np.int32), alive=np.zeros((B, A), np.uint8), imp=np.zeros((B, A), np.uint8),
                   jobpos=np.zeros((B, J, 2), np.int32), jobdone=np.zeros((B, J), np.uint8), used=np.zeros((B, A), np.uint8),
                   counts=np.zeros((B, A), np.int32), timer=np.zeros(B, np.int32), t=np.zeros(B, np.int32),
                   metrics=np.zeros((B, N_METRICS), np.int64), cursor=np.zeros(B, np.uint64))
        self.L.so_batch_export(self.envs, B, *[out[k].ctypes.data for k in
                               ("pos", "alive", "imp", "jobpos", "jobdone", "used", "counts", "timer", "t", "metrics", "cursor")])
        out["episode"] = self.episode.astype(np.uint32)
        return out

    def imp_idxs(self, b=0):
        return np.ctypeslib.as_array(self.envs[b].imp_idxs)[: self.envs[b].cfg.n_imposters].copy()

    def set_state(self, b, *, pos=None, alive=None, jobpos=None, jobdone=None, t=None, used=None, counts=None,
                  timer=None, imp_mask=None):
        e = self.envs[b]
        if pos is not None:
            for i, (x, y) in enumerate(np.asarray(pos)):
                e.pos[i][0], e.pos[i][1] = int(x), int(y)
        if alive is not None:
            for i, v in enumerate(alive):
                e.alive[i] = int(v)
        if jobpos is not None:
            for j, (x, y) in enumerate(np.asarray(jobpos).reshape(-1, 2)):
                e.jobpos[j][0], e.jobpos[j][1] = int(x), int(y)
        if jobdone is not None:
            for j, v in enumerate(jobdone):
                e.jobdone[j] = int(v)
        if t is not None:
            e.t = int(t)
        if used is not None:
            for i, v in enumerate(used):
                e.used[i] = int(v)
        if counts is not None:
            for i, v in enumerate(counts):
                e.counts[i] = int(v)
        if timer is not None:
            e.timer = int(timer)
        if imp_mask is not None:
            k = 0
            for i, v in enumerate(imp_mask):
                e.imp_mask[i] = int(v)
                if int(v):
                    e.imp_idxs[k] = i
                    k += 1
            for i in range(self.A):
                e.n_role_actions[i] = (6 if e.imp_mask[i] else 5) if e.cfg.variant == 1 else (7 if e.imp_mask[i] else 6)

    # -- observations -----------------------------------------------------------------------------
    def obs_raw_u8(self) -> np.ndarray:
        n = self.L.so_obs_raw_size(C.byref(self.envs[0]))
        out = np.zeros((self.B, n), dtype=np.uint8)
        self.L.so_batch_obs_raw(self.envs, self.B, out.ctypes.data)
        return out

    def obs_raw(self) -> np.ndarray:
        n = self.L.so_obs_raw_size(C.byref(self.envs[0]))
        out = np.zeros((self.B, n), dtype=np.float64)
        for b in range(self.B):
            self.L.so_obs_raw(C.byref(self.envs[b]), out[b].ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def obs_flat(self, components, envs=None) -> np.ndarray:
        """FlatFeaturizer rows of every env (or of the env indices `envs`: big batches check a sample)."""
        comp = np.array([FLAT[c] if isinstance(c, str) else int(c) for c in components], dtype=np.int32)
        cp = comp.ctypes.data_as(C.POINTER(C.c_int32))
        n = self.L.so_obs_flat_size(C.byref(self.envs[0]), cp, len(comp))
        if n < 0:
            raise ValueError("unknown flat component")
        idx = range(self.B) if envs is None else [int(b) for b in envs]
        out = np.zeros((len(idx), n), dtype=np.float32)
        for k, b in enumerate(idx):
            rc = self.L.so_obs_flat(C.byref(self.envs[b]), cp, len(comp), out[k].ctypes.data_as(C.POINTER(C.c_float)))
            if rc < 0:
                raise ValueError(f"flat component not applicable to this config (rc={rc})")
        return out

    def obs_planes(self):
        ns = self.A + self.J + (self.A if self.variant == 2 else 0)
        sp = np.zeros((self.B, self.A + 2, self.N, self.N), dtype=np.float32)
        non = np.zeros((self.B, ns), dtype=np.float32)
        for b in range(self.B):
            self.L.so_obs_planes(C.byref(self.envs[b]), sp[b].ctypes.data_as(C.POINTER(C.c_float)),
                                 non[b].ctypes.data_as(C.POINTER(C.c_float)))
        return sp, non


def mt19937_words(seed: int, n: int) -> np.ndarray:
    """First n raw 32-bit outputs of numpy's legacy stream after np.random.seed(seed) (via the oracle)."""
    ob = OracleBatch(make_config("itg", n_crew=1), 1)
    ob.seed_mt([seed])
    return ob.raw_words(n)[0]


def philox4x32_10(ctr, key) -> list:
    """Raw Philox4x32-10 block (for known-answer tests)."""
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().so_philox4x32_10(c, k, o)
    return list(o)
