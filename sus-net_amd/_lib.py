"""ctypes binding of the C ABI in include/susnet.h (libsusnet_hip.so).

The library is the product: there is NO fallback.  If it is missing or does not export every symbol the
header declares, importing this module's `lib()` raises -- loudly.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libsusnet_hip.so"
LIB_PATH = os.environ.get("SUSNET_LIB_PATH", os.path.join(PKG_DIR, LIB_NAME))  # override: A/B experiments only

ABI_VERSION = 6
MAX_AGENTS, MAX_JOBS, MAX_GRID, N_METRICS, N_LIFETIME = 16, 16, 16, 13, 12

VARIANT_BASE, VARIANT_ITG, VARIANT_TAGGING = 0, 1, 2
RNG_TAPE, RNG_PHILOX = 1, 2
U8, I32, I64, F32, F64 = 0, 1, 2, 3, 4
LAYOUT_AB, LAYOUT_BA = 0, 1
OBS_NONE, OBS_RAW, OBS_FLAT, OBS_PLANES, OBS_PERSP = 0, 1, 2, 3, 4
E_INVALID, E_HIP, E_STATE, E_ACTION_ASSERT, E_ACTION_INDEX, E_TAPE, E_ROW = -1, -2, -3, -4, -5, -6, -7

RECORD_DEFAULT, RECORD_COMPACT = 0, 1
FLAT_COMPONENTS = {"onehot_pos": 0, "coord_pos": 1, "alive_crew": 2, "l1_crew": 3, "closest_crew": 4,
                   "walls3x3": 5, "dist_to_imp": 6, "room_loc": 7}
LIFETIME_NAMES = ["episodes", "crew_won", "imposter_won", "truncated", "imp_killed_crew", "completed_jobs",
                  "sabotaged_jobs", "imp_voted_out", "crew_voted_out", "episode_steps", "env_steps", "reserved"]

# every symbol include/susnet.h declares
EXPORTS = [
    "susnet_abi_version", "susnet_last_error", "susnet_create", "susnet_destroy", "susnet_get_layout",
    "susnet_bind_state", "susnet_bind_tape", "susnet_seed", "susnet_tick", "susnet_reset", "susnet_sample_actions", "susnet_policy_actions", "susnet_qnet_packed_floats", "susnet_qnet_pack", "susnet_qnet_forward", "susnet_step", "susnet_policy_step", "susnet_qnet_policy_step", "susnet_qnet_policy_rollout",
    "susnet_rollout", "susnet_record_layout", "susnet_record_layout_of", "susnet_set_launch_limit", "susnet_observe", "susnet_obs_size", "susnet_featurize", "susnet_export_state", "susnet_import_state",
    "susnet_reduce_lifetime", "susnet_device_tick", "susnet_poll_errors", "susnet_ring_append", "susnet_scent",
]


class Config(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_uint32), ("abi_version", C.c_uint32), ("variant", C.c_int32), ("batch", C.c_int32),
        ("n_imposters", C.c_int32), ("n_crew", C.c_int32), ("n_jobs", C.c_int32), ("grid_n", C.c_int32),
        ("grid_rows", C.c_uint16 * MAX_GRID),
        ("kill_reward", C.c_double), ("complete_job_reward", C.c_double), ("sabotage_reward", C.c_double),
        ("time_step_reward", C.c_double), ("game_end_reward", C.c_double), ("dead_penalty", C.c_double),
        ("vote_reward", C.c_double),
        ("max_time_steps", C.c_int32), ("is_action_order_random", C.c_int32), ("shuffle_imposter_index", C.c_int32),
        ("tag_reset_interval", C.c_int32), ("auto_reset", C.c_int32), ("rng_mode", C.c_int32),
        ("seed", C.c_uint64), ("env_id_base", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32),
    ]


class Layout(C.Structure):
    _fields_ = [
        ("state_bytes", C.c_uint64), ("state_align", C.c_uint64), ("batch_padded", C.c_int32), ("n_agents", C.c_int32),
        ("n_actions_imposter", C.c_int32), ("n_actions_crew", C.c_int32), ("action_space_n", C.c_int32),
        ("obs_raw_size", C.c_int32), ("envs_per_wave", C.c_int32), ("test_overrides", C.c_uint32),
    ]


class ObsSpec(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("dtype", C.c_int32), ("n_components", C.c_int32), ("components", C.c_int32 * 16),
        ("out", C.c_void_p), ("out2", C.c_void_p),
    ]


class StepIO(C.Structure):
    _fields_ = [
        ("actions", C.c_void_p), ("actions_dtype", C.c_int32), ("actions_layout", C.c_int32),
        ("rewards", C.c_void_p), ("rewards_dtype", C.c_int32), ("rewards_layout", C.c_int32),
        ("done", C.c_void_p), ("truncated", C.c_void_p), ("obs", C.POINTER(ObsSpec)), ("term_obs", C.c_void_p), ("roles", C.c_void_p),
    ]


class RolloutIO(C.Structure):
    _fields_ = [
        ("n_ticks", C.c_int32), ("actions", C.c_void_p), ("rewards", C.c_void_p), ("done", C.c_void_p),
        ("truncated", C.c_void_p), ("obs", C.POINTER(ObsSpec)), ("record", C.c_void_p),
        ("term_obs", C.c_void_p), ("roles", C.c_void_p), ("record_format", C.c_int32),
    ]


class RingIO(C.Structure):
    _fields_ = [
        ("n_ticks", C.c_int32), ("trajectory_size", C.c_int32), ("actions", C.c_void_p), ("rewards", C.c_void_p),
        ("done", C.c_void_p), ("truncated", C.c_void_p), ("obs", C.c_void_p), ("term_obs", C.c_void_p), ("roles", C.c_void_p),
        ("window", C.c_void_p), ("max_size", C.c_int64), ("idx", C.c_int64), ("states", C.c_void_p), ("next_states", C.c_void_p),
        ("ring_actions", C.c_void_p), ("ring_rewards", C.c_void_p), ("ring_dones", C.c_void_p), ("ring_imposters", C.c_void_p),
        ("record", C.c_void_p), ("record_format", C.c_int32),
    ]


class FeedIO(C.Structure):
    _fields_ = [("actions", C.c_void_p), ("rewards", C.c_void_p), ("done", C.c_void_p), ("truncated", C.c_void_p), ("obs", C.c_void_p),
                ("term_obs", C.c_void_p), ("roles", C.c_void_p), ("q", C.c_void_p)]


class PolicyOpts(C.Structure):
    _fields_ = [("epsilon", C.c_float), ("mask_dead", C.c_int32), ("crew_packed", C.c_void_p), ("crew_dims", C.POINTER(C.c_int32)),
                ("crew_n_dims", C.c_int32), ("pad_", C.c_int32), ("crew_q_out", C.c_void_p)]


class RecordLayout(C.Structure):
    _fields_ = [("record_bytes", C.c_int32), ("off_rewards", C.c_int32), ("off_actions", C.c_int32), ("off_done", C.c_int32),
                ("off_truncated", C.c_int32), ("off_obs", C.c_int32), ("planar", C.c_int32), ("n_obs_segments", C.c_int32),
                ("obs_segments", (C.c_int32 * 2) * 4), ("flags_packed", C.c_int32)]


class StateView(C.Structure):
    _fields_ = [
        ("agent_positions", C.c_void_p), ("alive_agents", C.c_void_p), ("imposter_mask", C.c_void_p),
        ("job_positions", C.c_void_p), ("completed_jobs", C.c_void_p), ("used_tag_actions", C.c_void_p),
        ("tag_counts", C.c_void_p), ("tag_reset_timer", C.c_void_p), ("t", C.c_void_p), ("metrics", C.c_void_p),
        ("rng_cursor", C.c_void_p), ("lifetime", C.c_void_p), ("episode_index", C.c_void_p),
    ]


class SusnetError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsusnet_hip: {msg} (code {code})")
        self.code = code


_lib = None


def _assert_single_hip_runtime():
    """torch ships its own libamdhip64; ours must resolve to the SAME loaded runtime or stream handles and
    device pointers would belong to different runtimes."""
    seen = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    seen.add(line.split()[-1])
    except OSError:
        return
    if len(seen) > 1:
        raise ImportError(f"two HIP runtimes mapped in one process: {sorted(seen)}; import torch before sus-net_amd")


def lib():
    """Load libsusnet_hip.so (built by `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is the product and there is no CPU fallback. "
            "Build it with `python -m build_hip` or `__graft_entry__.build()` (needs hipcc, gfx950).")
    import torch  # noqa: F401  (loads torch's libamdhip64 first so ours binds to the same runtime)

    L = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing}")
    _assert_single_hip_runtime()
    P = C.POINTER
    L.susnet_abi_version.restype = C.c_int
    L.susnet_last_error.restype = C.c_char_p
    L.susnet_create.argtypes = [P(Config), P(C.c_void_p)]
    L.susnet_destroy.argtypes = [C.c_void_p]
    L.susnet_destroy.restype = None
    L.susnet_get_layout.argtypes = [C.c_void_p, P(Layout)]
    L.susnet_bind_state.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.susnet_bind_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.susnet_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.susnet_tick.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64), C.c_void_p]
    L.susnet_reset.argtypes = [C.c_void_p, C.c_void_p, P(ObsSpec), C.c_void_p]
    L.susnet_sample_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.susnet_step.argtypes = [C.c_void_p, P(StepIO), C.c_void_p]
    L.susnet_rollout.argtypes = [C.c_void_p, P(RolloutIO), C.c_void_p]
    L.susnet_policy_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, P(PolicyOpts), P(StepIO), C.c_void_p]
    L.susnet_qnet_policy_step.argtypes = [C.c_void_p, P(C.c_int32), C.c_int32, P(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p, P(PolicyOpts), P(StepIO),
                                          C.c_void_p]
    L.susnet_qnet_policy_rollout.argtypes = [C.c_void_p, P(C.c_int32), C.c_int32, P(C.c_int32), C.c_int32, C.c_void_p, P(PolicyOpts), P(FeedIO), C.c_int32,
                                             C.c_void_p]
    L.susnet_record_layout_of.argtypes = [C.c_void_p, C.c_int32, P(RecordLayout)]
    L.susnet_policy_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, P(PolicyOpts), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.susnet_qnet_packed_floats.argtypes = [C.c_void_p, P(C.c_int32), C.c_int32, P(C.c_int32), C.c_int32]
    L.susnet_qnet_pack.argtypes = [C.c_void_p, P(C.c_int32), C.c_int32, P(C.c_int32), C.c_int32, P(C.c_void_p), P(C.c_void_p), C.c_void_p, C.c_void_p]
    L.susnet_qnet_forward.argtypes = [C.c_void_p, P(C.c_int32), C.c_int32, P(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.susnet_record_layout.argtypes = [C.c_void_p, P(RecordLayout)]
    L.susnet_set_launch_limit.argtypes = [C.c_void_p, C.c_uint64]
    L.susnet_observe.argtypes = [C.c_void_p, P(ObsSpec), C.c_void_p]
    L.susnet_obs_size.argtypes = [C.c_void_p, P(ObsSpec), P(C.c_int32), P(C.c_int32)]
    L.susnet_featurize.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, P(ObsSpec), C.c_void_p]
    L.susnet_export_state.argtypes = [C.c_void_p, P(StateView), C.c_void_p]
    L.susnet_import_state.argtypes = [C.c_void_p, P(StateView), C.c_void_p]
    L.susnet_reduce_lifetime.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.susnet_device_tick.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.susnet_poll_errors.argtypes = [C.c_void_p, P(C.c_uint32), C.c_void_p]
    L.susnet_ring_append.argtypes = [C.c_void_p, P(RingIO), C.c_void_p]
    L.susnet_scent.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    for name in EXPORTS:
        if name not in ("susnet_last_error", "susnet_destroy"):
            getattr(L, name).restype = C.c_int
    L.susnet_qnet_packed_floats.restype = C.c_int64
    if L.susnet_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI mismatch: library {L.susnet_abi_version()} vs binding {ABI_VERSION}")
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise SusnetError(rc, lib().susnet_last_error().decode())
