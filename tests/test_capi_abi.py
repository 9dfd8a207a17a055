"""CPU-only checks of the C ABI boundary: the header compiles as plain C, the ctypes mirror has the same
struct layouts, the library loads and exports every declared symbol, and the host-side validation of
`susnet_create` mirrors the reference constructors' asserts (no kernel is launched here)."""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("sus-net_amd")


def test_header_is_plain_c_and_matches_ctypes(pkg, tmp_path):
    L = pkg._lib
    prog = tmp_path / "sizes.c"
    structs = {"susnet_config": L.Config, "susnet_layout": L.Layout, "susnet_obs_spec": L.ObsSpec,
               "susnet_step_io": L.StepIO, "susnet_rollout_io": L.RolloutIO, "susnet_state_view": L.StateView,
               "susnet_record_layout_t": L.RecordLayout, "susnet_ring_io": L.RingIO, "susnet_policy_opts": L.PolicyOpts,
               "susnet_feed_io": L.FeedIO}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "susnet.h"', "int main(void){"]
    for name, ct in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ["return 0;}"]
    prog.write_text("\n".join(lines))
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, ct in structs.items():
        assert int(out[name]) == C.sizeof(ct), name
        for fname, _ in ct._fields_:
            assert int(out[f"{name}.{fname}"]) == getattr(ct, fname).offset, f"{name}.{fname}"


def test_library_exports_every_declared_symbol(pkg):
    header = open(os.path.join(ROOT, "include", "susnet.h")).read()
    declared = set(re.findall(r"\b(susnet_[a-z_]+)\s*\(", header))
    assert declared == set(pkg._lib.EXPORTS), declared ^ set(pkg._lib.EXPORTS)
    lib = pkg._lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.susnet_abi_version() == pkg._lib.ABI_VERSION


def test_missing_library_fails_loudly(pkg, monkeypatch):
    L = pkg._lib
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libsusnet_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        L.lib()


def make_cfg(L, **kw):
    cfg = L.Config()
    cfg.struct_bytes, cfg.abi_version = C.sizeof(L.Config), L.ABI_VERSION
    d = dict(variant=L.VARIANT_BASE, batch=8, n_imposters=1, n_crew=2, n_jobs=4, grid_n=9, max_time_steps=1000,
             is_action_order_random=1, shuffle_imposter_index=1, tag_reset_interval=50, rng_mode=L.RNG_PHILOX)
    d.update(kw)
    for k, v in d.items():
        setattr(cfg, k, v)
    for i in range(min(cfg.grid_n, 16)):
        cfg.grid_rows[i] = ((1 << cfg.grid_n) - 1) & 0xFFFF
    return cfg


def test_create_validates_like_the_reference_constructors(pkg):
    L = pkg._lib
    lib = L.lib()
    h = C.c_void_p()

    def create(**kw):
        return lib.susnet_create(C.byref(make_cfg(L, **kw)), C.byref(h))

    assert create() == 0
    lay = L.Layout()
    assert lib.susnet_get_layout(h, C.byref(lay)) == 0
    # base.py:82-99,209: crew 6 actions, imposter 7, Discrete(8); flattened state 3A+3J (base.py:211-232)
    assert (lay.n_agents, lay.n_actions_crew, lay.n_actions_imposter, lay.action_space_n, lay.obs_raw_size) == (3, 6, 7, 8, 21)
    assert lay.state_bytes % 256 == 0 and lay.batch_padded == 256
    lib.susnet_destroy(h)
    # base.py:243-249
    assert create(n_imposters=0) == L.E_INVALID and b"imposter" in lib.susnet_last_error()
    assert create(n_crew=0) == L.E_INVALID
    assert create(n_jobs=-1) == L.E_INVALID
    assert create(n_imposters=2, n_crew=2) == L.E_INVALID and b"more crew" in lib.susnet_last_error().lower()
    # pred_prey.py:75-76: only n_crew > 0 is required; 1v1 allowed; n_imposters forced to 1
    assert create(variant=L.VARIANT_ITG, n_imposters=5, n_crew=1, n_jobs=0) == 0
    assert lib.susnet_get_layout(h, C.byref(lay)) == 0
    assert (lay.n_agents, lay.n_actions_crew, lay.n_actions_imposter, lay.obs_raw_size) == (2, 5, 6, 6)
    lib.susnet_destroy(h)
    # tagging.py:35-60: A-1 extra actions per agent, Discrete(8 + A), flattened state 3A+3J+2A+1
    assert create(variant=L.VARIANT_TAGGING, n_imposters=1, n_crew=4, n_jobs=5) == 0
    assert lib.susnet_get_layout(h, C.byref(lay)) == 0
    assert (lay.n_agents, lay.n_actions_crew, lay.n_actions_imposter, lay.action_space_n, lay.obs_raw_size) == (5, 10, 11, 13, 41)
    # calls that need device buffers are refused before any launch
    assert lib.susnet_reset(h, None, None, None) == L.E_STATE
    lib.susnet_destroy(h)
    # build limits
    assert create(n_crew=16) == L.E_INVALID
    assert create(grid_n=17) == L.E_INVALID
    assert create(struct_bytes=4) == L.E_INVALID
    assert create(rng_mode=0) == L.E_INVALID


def test_packed_record_layouts(pkg):
    """susnet_record_layout: compiled-in configurations only; fields inside the record, no overlap, dword-sized; the byte-parallel
    kernels put the raw row right behind the actions (job cells on a dword boundary), the 1v1 kernels done / truncated first."""
    L = pkg._lib
    lib = L.lib()
    h = C.c_void_p()
    lay = L.RecordLayout()

    def layout(**kw):
        assert lib.susnet_create(C.byref(make_cfg(L, **kw)), C.byref(h)) == 0
        info = L.Layout()
        assert lib.susnet_get_layout(h, C.byref(info)) == 0
        assert lib.susnet_record_layout(h, C.byref(lay)) == 0
        lib.susnet_destroy(h)
        return info.n_agents, info.obs_raw_size

    itg = dict(variant=L.VARIANT_ITG, n_crew=1, n_jobs=0, is_action_order_random=0, shuffle_imposter_index=0)
    cfg4 = dict(n_imposters=2, n_crew=6, n_jobs=4, grid_n=14)
    for kw, want_obs_first in ((itg, False), (dict(n_crew=2, grid_n=14), True), (dict(cfg4, batch=65536), True),
                               (dict(cfg4, batch=32768), True), (dict(variant=L.VARIANT_TAGGING, n_crew=4, n_jobs=5), True)):
        A, F = layout(**kw)
        assert lay.record_bytes > 0 and lay.record_bytes % 4 == 0, kw
        used = [False] * lay.record_bytes
        for off, n in ((lay.off_rewards, 4 * A), (lay.off_actions, A), (lay.off_done, 1), (lay.off_truncated, 1), (lay.off_obs, F)):
            assert 0 <= off and off + n <= lay.record_bytes and not any(used[off:off + n]), (kw, off, n)
            used[off:off + n] = [True] * n
        assert lay.off_rewards % 4 == 0
        assert (lay.off_obs < lay.off_done) == want_obs_first, kw
        if want_obs_first:  # the byte-parallel kernels: raw row right behind the actions, its job cells on a dword boundary
            assert lay.off_obs == 5 * A and (lay.off_obs + 3 * A) % 4 == 0
        assert lay.planar == (1 if want_obs_first else 0) and lay.n_obs_segments == 1 and tuple(lay.obs_segments[0]) == (lay.off_obs, F)
    # the byte-parallel family (job count at run time): a layout independent of the job count, the observation in segments
    sizes = set()
    for jobs in (0, 2, 5, 8):
        A, F = layout(n_imposters=1, n_crew=3, n_jobs=jobs)
        sizes.add(lay.record_bytes)
        assert lay.record_bytes > 0 and lay.record_bytes % 4 == 0 and lay.planar == 1
        segs = [tuple(lay.obs_segments[k]) for k in range(lay.n_obs_segments)]
        assert sum(n for _, n in segs) == F and segs[0] == (lay.off_obs, 3 * A) and lay.n_obs_segments == (1 if jobs == 0 else 3)
        used = [False] * lay.record_bytes
        for off, n in [(lay.off_rewards, 4 * A), (lay.off_actions, A), (lay.off_done, 1), (lay.off_truncated, 1)] + segs:
            assert 0 <= off and off + n <= lay.record_bytes and not any(used[off:off + n]), (jobs, off, n)
            used[off:off + n] = [True] * n
    assert len(sizes) == 1
    A, F = layout(variant=L.VARIANT_TAGGING, n_imposters=2, n_crew=5, n_jobs=3)
    assert lay.n_obs_segments == 4 and sum(lay.obs_segments[k][1] for k in range(4)) == F
    A, F = layout(n_imposters=3, n_crew=5, n_jobs=2)  # three imposters: in the family since round 5 (the per-turn SpecA<3..8> kernels are gone)
    assert lay.record_bytes > 0 and lay.planar == 1 and lay.n_obs_segments == 3
    A, F = layout(n_imposters=4, n_crew=9, n_jobs=2)  # more than 8 agents: the generic kernels, no packed mode
    assert lay.record_bytes == 0


def test_obs_sizes_follow_the_reference_featurizers(pkg):
    L = pkg._lib
    lib = L.lib()
    h = C.c_void_p()
    assert lib.susnet_create(C.byref(make_cfg(L, n_crew=2, grid_n=14)), C.byref(h)) == 0
    spec = L.ObsSpec()
    f1, f2 = C.c_int32(), C.c_int32()
    spec.mode, spec.dtype = L.OBS_PLANES, L.F32
    assert lib.susnet_obs_size(h, C.byref(spec), C.byref(f1), C.byref(f2)) == 0
    assert (f1.value, f2.value) == (5 * 14 * 14, 3 + 4)  # model_ready.py:230-247
    spec.mode, spec.n_components = L.OBS_FLAT, 3
    for i, c in enumerate(["onehot_pos", "alive_crew", "closest_crew"]):
        spec.components[i] = L.FLAT_COMPONENTS[c]
    assert lib.susnet_obs_size(h, C.byref(spec), C.byref(f1), C.byref(f2)) == 0
    assert f1.value == 3 * 28 + 2 + 2  # 84 + 2 + 2 = 88
    spec.components[0] = L.FLAT_COMPONENTS["room_loc"]  # 9x9 only (component.py:8-17)
    assert lib.susnet_obs_size(h, C.byref(spec), C.byref(f1), C.byref(f2)) == L.E_INVALID
    lib.susnet_destroy(h)


def test_qnet_pack_layout(pkg):
    """susnet_qnet_pack (host code): the packed image of a reference MLP [88, 200, 100, 50, 16, 7] -- hidden widths that need padding --
    against the layout include/susnet.h / susnet_qnet.h describe: layer 1 transposed (one padded row per position bit, a zero row, and
    one row per combination of the alive / closest bits = b1 + their columns), layers 2.. as 32 x 32 blocks in [k block][row block] order whose lane l holds row 32 nb + l % 32 and, as four float4, the columns
    32 kb + 8 q + 4 (l / 32) + r; unserved shapes are refused."""
    L = pkg._lib
    lib = L.lib()
    h = C.c_void_p()
    assert lib.susnet_create(C.byref(make_cfg(L, n_crew=2, grid_n=14)), C.byref(h)) == 0
    comps = (C.c_int32 * 3)(*[L.FLAT_COMPONENTS[k] for k in ("onehot_pos", "alive_crew", "closest_crew")])
    dims = [88, 200, 100, 50, 16, 7]
    cd = (C.c_int32 * 6)(*dims)
    n = lib.susnet_qnet_packed_floats(h, comps, 3, cd, 6)
    pad = [88, 256, 128, 64, 32, 32]
    RS = 256 + 4
    rows = 84 + 1 + 16  # position bits, the zero row, 2^4 combinations of the four bits behind the one-hots
    w1_floats = -(-rows * RS // 1024) * 1024  # (zero-padded to whole rounds of 4 KiB: the both-teams tick swaps it by 1 KiB transfers, one per wave and round)
    lds = w1_floats + sum(pad[2:])  # what a workgroup copies to LDS: layer 1 transposed, then the biases of layers 2..5
    assert n == lds + sum(pad[l] * pad[l + 1] for l in range(1, 5)) + 4
    rng = np.random.default_rng(0)
    W = [rng.standard_normal((dims[l + 1], dims[l])).astype(np.float32) for l in range(5)]
    Bv = [rng.standard_normal(dims[l + 1]).astype(np.float32) for l in range(5)]
    sl = rng.random(4).astype(np.float32)
    out = np.full(n, np.nan, dtype=np.float32)
    wp = (C.c_void_p * 5)(*[w.ctypes.data for w in W])
    bp = (C.c_void_p * 5)(*[b.ctypes.data for b in Bv])
    assert lib.susnet_qnet_pack(h, comps, 3, cd, 6, wp, bp, sl.ctypes.data, out.ctypes.data) == 0
    w1 = out[:rows * RS].reshape(rows, RS)
    np.testing.assert_array_equal(w1[:84, :200], W[0].T[:84])
    assert not w1[:, 200:].any() and not w1[84].any()
    for v in range(16):
        want = Bv[0].copy()
        for bit in range(4):
            if (v >> bit) & 1:
                want = want + W[0][:, 84 + bit]  # float32, lowest bit first
        np.testing.assert_array_equal(w1[85 + v, :200], want)
    assert not out[rows * RS:w1_floats].any()
    off = w1_floats
    for l in range(1, 5):
        bias = np.zeros(pad[l + 1], dtype=np.float32)
        bias[:dims[l + 1]] = Bv[l]
        np.testing.assert_array_equal(out[off:off + pad[l + 1]], bias)
        off += pad[l + 1]
    assert off == lds
    for l in range(1, 5):  # the weight stream: layers 2..5 back to back, 32 x 32 blocks in [k block][row block] order
        KP, NP = pad[l], pad[l + 1]
        full = np.zeros((NP, KP), dtype=np.float32)
        full[:dims[l + 1], :dims[l]] = W[l]
        blk = out[off:off + KP * NP].reshape(KP // 32, NP // 32, 4, 64, 4)
        kb, nb, q, lane, r = np.meshgrid(*[np.arange(k) for k in blk.shape], indexing="ij")
        np.testing.assert_array_equal(blk, full[32 * nb + lane % 32, 32 * kb + 8 * q + 4 * (lane // 32) + r])
        off += KP * NP
    np.testing.assert_array_equal(out[off:], sl)
    # refused: another depth, a layer wider than the compiled-in family, a feature layout without a compiled-in writer
    assert lib.susnet_qnet_packed_floats(h, comps, 3, (C.c_int32 * 5)(88, 256, 128, 64, 7), 5) == L.E_INVALID
    assert lib.susnet_qnet_packed_floats(h, comps, 3, (C.c_int32 * 6)(88, 512, 128, 64, 16, 7), 6) == L.E_INVALID
    assert lib.susnet_qnet_packed_floats(h, comps, 1, cd, 6) == L.E_INVALID
    lib.susnet_destroy(h)


@pytest.fixture(scope="module")
def isa(pkg):
    """sus-net_amd/isa_checks.py on the shipped library.  The LLVM tools ship with hipcc: where the library could be built
    they are there, and their absence fails the checks instead of skipping them."""
    mod = importlib.import_module("sus-net_amd.isa_checks")
    assert mod.tools_available(), "llvm-objdump / llvm-readelf / c++filt not found next to hipcc"
    return mod


@pytest.fixture(scope="module")
def shipped(pkg, isa):
    """(per-kernel rows, problems) of the shipped library: disassembled once for the module"""
    return isa.verify_library(pkg._lib.LIB_PATH)


def test_shipped_library_passes_the_build_time_isa_checks(shipped):
    """What build_hip.build() enforces after every link: no scratch traffic (per-lane arrays indexed at run time get demoted to
    private memory: slow, and every scratch load carries an `s_waitcnt vmcnt(0)` that also waits for all outstanding trajectory
    stores), no vector write into a wide store's data registers within two wait states, the register budget of the
    benchmarked kernels, no static LDS under the table area."""
    rows, problems = shipped
    assert not problems, problems
    assert len(rows) > 50 and sum(r.get("valu", 0) for r in rows) > 100000, "disassembly looks empty"


def test_headline_kernels_keep_their_register_budget(isa, shipped):
    """the four packed-record rollout kernels bench.py times spill at most 16 SGPRs, use no accumulation
    registers, no `v_accvgpr` / scratch moves, and fit the instruction cache (the round-2 kernels unrolled whole tick groups:
    36 000 instructions, 255 spilled SGPRs, 256 VGPRs + 73 AGPRs)."""
    rows = {r["name"]: r for r in shipped[0]}
    for k in isa.HEADLINE_KERNELS:
        r = rows[k]
        assert r["sgpr_spill"] <= 16 and r["agpr"] == 0 and r["vgpr_spill"] == 0 and r["v_accvgpr"] == 0 and r["scratch"] == 0, (k, r)
        assert r["v_readlane"] + r["v_writelane"] <= 2 * 16, (k, r)  # spill moves: one write + (at most a few) reads per spilled SGPR
        assert r["instructions"] < 6000 and r["code_bytes"] < 48 * 1024, (k, r)
    # a deliberately broken budget is reported
    worse = [dict(r, sgpr_spill=17) if r["name"] == isa.HEADLINE_KERNELS[0] else r for r in rows.values()]
    assert any("sgpr_spill" in p for p in isa.check_limits(worse))


def test_store_data_hazard_checker_flags_the_measured_case(isa):
    asm = "\n".join(
        [
            "\tbuffer_store_dwordx4 v[178:181], v91, s[76:79], s6 offen   // 0001: E07C1000",
            "\tv_alignbit_b32 v179, v53, v52, 24   // 0002: D1CE00B3",
        ]
    )
    assert len(isa.store_data_hazards(asm)) == 1
    ok = asm.replace("\tv_alignbit", "\ts_nop 1\n\tv_alignbit")
    assert isa.store_data_hazards(ok) == []
    assert isa.store_data_hazards("global_store_dwordx4 v[2:3], v[4:7], off\nv_lshl_add_u64 v[2:3], v[2:3], 0, s[16:17]") == []
    assert len(isa.store_data_hazards("global_store_dwordx4 v[2:3], v[4:7], off\nv_mov_b32_e32 v7, 0")) == 1
    assert isa.scratch_instructions("\tscratch_load_dword v1, off, s32\n\tv_mov_b32 v0, v1") == ["scratch_load_dword v1, off, s32"]


def _listing(name, lines):
    """A disassembly listing as llvm-objdump prints it, from (label or None, instruction) pairs; `@label` in a branch becomes its offset."""
    addr, at, sized = 0x1000, {}, []
    for label, text in lines:
        size = 8 if text.startswith(("v_accvgpr", "global_", "v_mfma", "v_readlane", "v_writelane")) else 4
        if label:
            at[label] = addr
        sized.append((addr, size, text))
        addr += size
    out = [f"{0x1000:016x} <{name}>:"]
    for a, size, text in sized:
        if "@" in text:
            op, lab = text.split("@")
            text = f"{op}{((at[lab] - (a + size)) // 4) & 0xffff}"
        out.append(f"\t{text:<60}// {a:012X}: " + " ".join(["BF800000"] * (size // 4)))
    return "\n".join(out)


def test_parking_move_checker_is_value_level(isa):
    """isa_checks.parked_under_divergence: an accumulator-register value read under a wider EXEC than it was written under is reported --
    the round-4 fault (k_qnet_step: the Q-row index parked inside `if (b + 32 < B)`, read after the region's end) -- and nothing else is:
    a partial update of a value whose home is the accumulator register, a read inside the region, a region entered through a mask that
    was spilled to a vector lane.  Every kernel is looked at, not a named few."""
    K = "_ZN6susnet14k_rollout_swarINS_4SpecILi8ELin1ELi2ELi1ELin1ELi2EEELi0ENS_9PhiloxRngEEEvNS_6ConstsE"
    region = lambda body, after: [
        (None, "s_and_saveexec_b64 s[12:13], vcc"),
        (None, "s_cbranch_execz @end"),
        *[(None, b) for b in body],
        ("end", "s_or_b64 exec, exec, s[12:13]"),
        *[(None, a) for a in after],
        (None, "s_endpgm"),
    ]
    # the measured case: written inside the region only, read after its end
    bad = isa.parked_under_divergence(_listing(K, region(["v_accvgpr_write_b32 a207, v11"], ["v_accvgpr_read_b32 v11, a207"])))
    assert [(b[1], b[2]) for b in bad] == [("v_accvgpr_write_b32 a207, v11", "v_accvgpr_read_b32 v11, a207")]
    # ... read back INSIDE the region: fine
    assert isa.parked_under_divergence(_listing(K, region(["v_accvgpr_write_b32 a207, v11", "v_accvgpr_read_b32 v11, a207"], []))) == []
    # a value that lives in the accumulator register (whole write first), updated for some lanes inside the region: fine
    whole = [(None, "v_accvgpr_write_b32 a1, v87")] + region(["v_accvgpr_write_b32 a1, v88"], ["v_accvgpr_read_b32 v28, a1"])
    assert isa.parked_under_divergence(_listing(K, whole)) == []
    # the region's mask restored from a register the analysis cannot name: EXEC may be anything -> the narrow write is reported
    unknown = region(["v_accvgpr_write_b32 a3, v1"], ["v_accvgpr_read_b32 v1, a3"])
    unknown[3] = ("end", "s_or_b64 exec, exec, s[40:41]")
    assert len(isa.parked_under_divergence(_listing(K, unknown))) == 1
    # ... but a saved mask that went through a spill lane and came back under another name is followed
    spilled = [
        (None, "s_and_saveexec_b64 s[12:13], vcc"),
        (None, "v_writelane_b32 v254, s12, 3"),
        (None, "v_writelane_b32 v254, s13, 4"),
        (None, "s_cbranch_execz @end"),
        (None, "v_accvgpr_write_b32 a5, v1"),
        (None, "v_accvgpr_read_b32 v1, a5"),
        ("end", "v_readlane_b32 s40, v254, 3"),
        (None, "v_readlane_b32 s41, v254, 4"),
        (None, "s_or_b64 exec, exec, s[40:41]"),
        (None, "s_endpgm"),
    ]
    assert isa.parked_under_divergence(_listing(K, spilled)) == []
    # a loop: written inside an `if` of one iteration, read inside the `if` of the next (other lanes): reported
    loop = [
        (None, "v_accvgpr_write_b32 a9, v0"),
        ("top", "s_and_saveexec_b64 s[4:5], vcc"),
        (None, "s_cbranch_execz @skip"),
        (None, "v_accvgpr_read_b32 v2, a9"),
        (None, "v_accvgpr_write_b32 a9, v3"),
        ("skip", "s_or_b64 exec, exec, s[4:5]"),
        (None, "s_cbranch_scc1 @top"),
        (None, "s_endpgm"),
    ]
    assert isa.parked_under_divergence(_listing(K, loop)) == []  # (whole before the loop, partial updates inside: its home)
    loop[0] = (None, "s_nop 0")
    assert len(isa.parked_under_divergence(_listing(K, loop))) == 1
    # MFMA reads every lane of its accumulator operand whatever EXEC is
    mf = region(["v_accvgpr_write_b32 a0, v1", "v_mfma_f32_32x32x2_f32 a[0:15], v2, v3, a[0:15]"], [])
    assert len(isa.parked_under_divergence(_listing(K, mf))) >= 1
    # kernels can still be selected by name
    assert isa.parked_under_divergence(_listing(K, region(["v_accvgpr_write_b32 a207, v11"], ["v_accvgpr_read_b32 v11, a207"])), kernels=("k_qnet",)) == []


def test_scheduling_strategy_per_translation_unit(pkg):
    """`-amdgpu-sched-strategy=max-ilp` goes to the stepping units that run at one wave per SIMD and gain from it (cfg3 / cfg4 / tag5 / the
    family / the generic kernels) and to nothing else: the Q-network units (hand-scheduled matrix sections: 11 % slower), the 1v1 unit (its
    wall-map kernel: 3 % slower) and susnet_capi.hip (many-waves kernels that live on occupancy) keep the default -- build_hip.flags_for."""
    import importlib

    bh = importlib.import_module(pkg.__name__ + ".build_hip")
    ilp = lambda name: "-amdgpu-sched-strategy=max-ilp" in bh.flags_for("/x/" + name)
    assert all(ilp(n) for n in ("inst_cfg3.hip", "inst_cfg4.hip", "inst_tag5.hip", "inst_fam_a8_tag.hip", "inst_fam_a12_base_ni3.hip", "inst_generic.hip", "inst_a2.hip"))
    assert not any(ilp(n) for n in ("susnet_capi.hip", "inst_cfg2.hip", "inst_qnet_onehot1.hip", "inst_qnet_onehot3.hip", "inst_qnet_coord1.hip"))
    import os

    have = {os.path.basename(f) for f in bh.sources()}
    assert {"susnet_capi.hip", "inst_cfg2.hip", "inst_cfg3.hip", "inst_cfg4.hip", "inst_tag5.hip", "inst_qnet_onehot3.hip"} <= have


def test_m0_is_only_written_by_the_lds_transfer_moves(isa, shipped):
    """The both-teams tick's image transfers set M0 from inline assembly without the compiler knowing (a reserved register cannot be
    clobbered): sound only while nothing else in the library touches M0 -- checked on the shipped ISA by every build (`m0_conflicts`)."""
    assert isa.m0_conflicts("\ts_mov_b32 m0, s13\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v0, s[10:11]\n") == []
    assert isa.m0_conflicts("\ts_mov_b32 m0, s13\n\ts_movrels_b32 s1, s2\n\tds_gws_init v0 gds\n\ts_mov_b32 m0, 0x1000\n\tv_readlane_b32 s1, v2, m0\n") == [
        "s_mov_b32 m0, 0x1000", "v_readlane_b32 s1, v2, m0"]
    assert not [p for p in shipped[1] if "M0" in p]


def test_no_accumulator_registers_outside_the_q_network_kernels(isa, shipped):
    """the value-level check passes on every kernel (test_shipped_library_passes_the_build_time_isa_checks), and the
    kernels that need no matrix core keep out of the accumulator registers altogether -- the environment kernels run under divergent
    regions by design, so they are kept below 256 vector registers instead of relying on the proof."""
    rows = shipped[0]
    users = sorted(r["name"] for r in rows if r.get("agpr", 0) > 0 or r.get("v_accvgpr", 0) > 0)
    assert users and all(n.startswith("k_qnet") for n in users), users


def test_build_rejects_a_library_that_fails_the_isa_checks(pkg, isa, tmp_path, monkeypatch):
    """build_hip.verify(): a library with a problem is moved aside and the build raises."""
    import shutil

    lib = tmp_path / "libsusnet_hip.so"
    shutil.copy(pkg._lib.LIB_PATH, lib)  # (that the shipped library passes: test_shipped_library_passes_the_build_time_isa_checks)
    monkeypatch.setattr(isa, "HEADLINE_LIMITS", dict(isa.HEADLINE_LIMITS, code_bytes=1024))
    with pytest.raises(RuntimeError, match="failed the ISA checks"):
        pkg.build_hip.verify(str(lib))
    assert not lib.exists() and (tmp_path / "libsusnet_hip.so.rejected").exists()
