#!/usr/bin/env python3
"""Checks on the SHIPPED gfx950 code objects (the ISA inside libsusnet_hip.so), run by build_hip.build() after every link
and by tests/test_capi_abi.py:

  * store_data_hazards(): no vector write into the data registers of a > 64-bit store within two wait states (a gfx950
    hazard the compiler only guards when the store's scalar-offset field is a constant: see BufDst::st128);
  * scratch_instructions(): no scratch (private memory) traffic;
  * parked_under_divergence(): in the kernels at the register limit (k_qnet*), no value parked in an accumulator register inside a divergent region;
  * collect(): per kernel the ELF notes (`llvm-readelf --notes`: vgpr / agpr / sgpr counts, spill counts, private segment,
    static LDS) and disassembly statistics (`llvm-objdump -d`: static instruction count; `v_readlane` / `v_writelane` = SGPR
    spills to vector lanes, `v_accvgpr_*` = VGPR spills to accumulation registers, `scratch_*` = memory);
  * check_limits(): the register budget of the kernels bench.py times (HEADLINE_LIMITS).

    python sus-net_amd/isa_checks.py [--match SUBSTR] [--json OUT] [--md OUT]      prints the table
"""
from __future__ import annotations

import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LLVM = os.environ.get("SUSNET_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
LIB = os.path.join(PKG_DIR, "libsusnet_hip.so")


CXXFILT = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")


def tools_available() -> bool:
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-readelf")) and os.path.exists(CXXFILT)


def code_objects(lib: str, workdir: str):
    so = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=workdir, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return sorted(os.path.join(workdir, p) for p in os.listdir(workdir) if "gfx950" in p)


def demangle(names):
    if not names:
        return {}
    out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return dict(zip(names, out.splitlines()))


def short_name(dem: str) -> str:
    s = dem.replace("susnet::", "").replace("void ", "")
    s = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", s)  # argument list
    s = s.replace("(susnet::._anon_0)", "").replace("(anonymous namespace)::", "")
    return s


NOTE_KEYS = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".sgpr_spill_count": "sgpr_spill",
             ".vgpr_spill_count": "vgpr_spill", ".private_segment_fixed_size": "scratch_bytes", ".group_segment_fixed_size": "lds_static",
             ".kernarg_segment_size": "kernarg_bytes"}


def parse_notes(obj: str):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], capture_output=True, text=True, check=True).stdout
    # the notes are YAML: a kernel entry of `amdhsa.kernels:` starts at "  - .key:" (indent 2), its own keys sit at indent 4,
    # everything deeper belongs to `.args`
    kernels, cur, inside = [], None, False
    for ln in txt.splitlines():
        if ln.startswith("amdhsa.kernels:"):
            inside = True
            continue
        if inside and ln and not ln.startswith(" "):
            inside, cur = False, None
        if not inside:
            continue
        m = re.match(r"^(  - |    )(\.[a-z_]+):\s*(.*)$", ln)
        if not m:
            continue
        if m.group(1) == "  - ":
            cur = {}
            kernels.append(cur)
        key, val = m.group(2), m.group(3).strip()
        if cur is None:
            continue
        if key == ".name":
            cur["mangled"] = val.strip("'\"")
        elif key in NOTE_KEYS:
            try:
                cur[NOTE_KEYS[key]] = int(val)
            except ValueError:
                pass
    return [k for k in kernels if "mangled" in k and "vgpr" in k]


def parse_disasm(obj: str, asm: str | None = None):
    """mangled kernel name -> static instruction statistics"""
    if asm is None:
        asm = disassemble(obj)
    stats, cur = {}, None
    for ln in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", ln)
        if m:
            cur = stats.setdefault(m.group(1), dict(instructions=0, code_bytes=0, v_readlane=0, v_writelane=0, v_accvgpr=0, scratch=0, s_nop=0,
                                                    s_waitcnt=0, ds=0, vmem_store=0, valu=0, salu=0, branch=0))
            continue
        if cur is None or "\t" not in ln:
            continue
        body = ln.split("//")
        ins = body[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        cur["instructions"] += 1
        if len(body) > 1:  # "// 000000001234: AABBCCDD EEFF0011"
            enc = [w for w in body[1].split(":")[-1].split() if re.fullmatch(r"[0-9A-Fa-f]{8}", w)]  # (a branch line ends in "<symbol+0x..>")
            cur["code_bytes"] += 4 * len(enc)
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            cur["v_readlane"] += op.startswith("v_readlane")
        if op.startswith("v_writelane"):
            cur["v_writelane"] += 1
        if op.startswith("v_accvgpr"):
            cur["v_accvgpr"] += 1
        if op.startswith("scratch_"):
            cur["scratch"] += 1
        if op == "s_nop":
            cur["s_nop"] += 1
        if op == "s_waitcnt":
            cur["s_waitcnt"] += 1
        if op.startswith("ds_"):
            cur["ds"] += 1
        if re.match(r"^(buffer|global|flat)_store", op):
            cur["vmem_store"] += 1
        if op.startswith("v_"):
            cur["valu"] += 1
        if op.startswith("s_") and not op.startswith(("s_cbranch", "s_branch", "s_nop", "s_waitcnt")):
            cur["salu"] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur["branch"] += 1
    return stats


def disassemble(obj: str) -> str:
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", obj], capture_output=True, text=True, check=True).stdout


def _vgprs(operand):
    """'v12' -> {12}, 'v[4:7]' -> {4..7}, anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def store_data_hazards(asm, window=2):
    """Vector writes to the data registers of a > 64-bit store within `window` wait states after it.

    The store reads its data over several cycles; gfx950 needs two wait states before a VALU instruction may overwrite
    them (measured: record dwords of lanes 12-15 / 28-31 carried the next tick's values).  The compiler inserts the
    `s_nop` only when the store's scalar-offset field is a constant, which is why BufDst::st128 keeps it 0.
    Validated against: AMD clang 22.0.0git (ROCm 7.2.0)."""
    lines = []
    for ln in asm.splitlines():
        ln = ln.split("//")[0].strip()
        if ln and not ln.endswith(":") and not ln.startswith((".", ";", "/")):
            lines.append(ln)
    # data operand: first for buffer stores, second (after the address) for global / flat stores
    wide = re.compile(r"^(?:buffer_store_(?:dwordx[34]|format_xyzw?)\s+|(?:global|flat)_store_dwordx[34]\s+[^,]+,\s*)(v\[\d+:\d+\])")
    bad = []
    for i, ln in enumerate(lines):
        m = wide.match(ln)
        if not m:
            continue
        data = _vgprs(m.group(1))
        waited = 0
        for nxt in lines[i + 1 : i + 1 + window]:
            if waited >= window or nxt.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                break
            mn = re.match(r"^s_nop\s+(\d+)", nxt)
            if mn:
                waited += int(mn.group(1)) + 1
                continue
            if nxt.startswith("v_"):
                ops = nxt.split(None, 1)[1].split(",") if " " in nxt else []
                if ops and _vgprs(ops[0].strip()) & data:
                    bad.append((ln, nxt))
            waited += 1
    return bad


# ---- accumulator registers and divergence: a value-level check ---------------------------------------------------------------------
# At the register limit the allocator keeps values in accumulator registers (`v_accvgpr_write_b32 aN, vM` ... `v_accvgpr_read_b32 vK, aN`).
# Such a copy only moves the lanes of the CURRENT EXEC.  A write under a narrowed EXEC whose value is read back under a wider one hands the
# lanes in between whatever the accumulator held before (round 4: k_qnet_step's last, partial wave stored Q rows through a wild pointer --
# the row index had been parked inside `if (b + 32 < B)` and was read after the region's end; profiles/r04_parking_fault.md).
#
# parked_under_divergence() proves, for every accumulator value of a kernel, that EXEC at each of its reads is a subset of EXEC at the
# write that produced it -- or reports the (write, read) pair.  Method:
#   1. EXEC versions.  Every instruction that writes EXEC defines a version; version 0 is the kernel's entry mask; `parent[v]` is a
#      version known to be a SUPERSET of v.  Narrowing forms (s_and / s_andn2 / s_*_saveexec with exec as an operand, the `else` xor) give
#      a child of the current version.  Widening forms (`s_or_b64 exec, exec, sX`, `s_mov_b64 exec, sX`, `s_or_saveexec_b64`) give the version
#      sX is KNOWN to hold lanes of -- scalar registers are tracked symbolically through s_mov / s_xor / s_andn2 / s_or and through their
#      spills to vector lanes (v_writelane / v_readlane) -- or, when sX is unknown, a fresh version only known to lie inside version 0.
#      Where control flow joins with different versions a phi version below their common ancestor is made.  (Assumed of the compiler: a
#      restore from a register that holds lanes of version u brings back exactly u -- structured control flow lowering does.)
#   2. Reaching definitions of the accumulator registers over the control-flow graph.  A definition carries the version it was made
#      under and an `exposed` flag, set as soon as it flows through an EXEC write (or a join) whose new version is not provably inside
#      its own.  MFMA reads and writes all lanes whatever EXEC is: its accumulator operands count as read / written under version 0.
#   3. A read is reported when a reaching definition is exposed, or was made under a version that is not an ancestor of the read's.
# Validated against: AMD clang 22.0.0git (ROCm 7.2.0).
_SREG = re.compile(r"^s(\d+)$")
_SRANGE = re.compile(r"^s\[(\d+):(\d+)\]$")
_AREG = re.compile(r"^a(\d+)$")
_ARANGE = re.compile(r"^a\[(\d+):(\d+)\]$")
_VCC = {"vcc": (106, 107), "vcc_lo": (106,), "vcc_hi": (107,)}
_NO_SDST = ("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_setreg", "s_store",
            "s_buffer_store", "s_dcache", "s_icache", "s_sethalt", "s_trap", "s_setpc", "s_sendmsg", "s_setkill", "s_inst_prefetch", "s_clause", "s_set_gpr",
            "s_ttrace", "s_decperflevel", "s_incperflevel", "s_cmpk", "s_setvskip", "s_code_end")


def _sregs(operand):
    m = _SREG.match(operand)
    if m:
        return (int(m.group(1)),)
    m = _SRANGE.match(operand)
    if m:
        return tuple(range(int(m.group(1)), int(m.group(2)) + 1))
    return _VCC.get(operand, ())


def _aregs(operand):
    m = _AREG.match(operand)
    if m:
        return (int(m.group(1)),)
    m = _ARANGE.match(operand)
    if m:
        return tuple(range(int(m.group(1)), int(m.group(2)) + 1))
    return ()


def _parse_kernels(asm, want):
    """-> [(mangled name, [(address, size, op, [operands])])] for the kernels `want(name, body_text)` selects."""
    out, cur, ins, raw = [], None, [], []

    def flush():
        if cur is not None and ins and want(cur, raw):
            out.append((cur, list(ins)))

    for ln in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", ln)
        if m:
            flush()
            cur, ins, raw = m.group(1), [], []
            continue
        if cur is None or "\t" not in ln or "//" not in ln:
            continue
        text, tail = ln.split("//", 1)
        text = text.strip()
        mm = re.match(r"\s*([0-9A-Fa-f]+):\s*(.*)$", tail)
        if not text or not mm:
            continue
        parts = text.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        # (modifiers ride on the last operand: "v153 offset:39520", "off offset:64")
        ops = [o.split()[0] if o and " " in o and not o.startswith(("s[", "v[", "a[")) else o for o in ops]
        size = 4 * sum(1 for w in mm.group(2).split() if re.fullmatch(r"[0-9A-Fa-f]{8}", w))  # (a branch line ends in "<symbol+0x..>")
        ins.append((int(mm.group(1), 16), size, parts[0], ops))
        raw.append(parts[0])
    flush()
    return out


class _ExecVersions:
    """parent[] forest of EXEC versions (see above); ids are tied to instruction sites so that the data flow converges."""

    def __init__(self):
        self.parent = {0: None}
        self.ids = {}
        self.changed = False

    def depth_path(self, v):
        path = []
        while v is not None:
            path.append(v)
            v = self.parent[v]
        return path

    def anc_or_eq(self, a, v):
        while v is not None:
            if v == a:
                return True
            v = self.parent[v]
        return False

    def lca(self, a, b):
        pa = set(self.depth_path(a))
        while b not in pa:
            b = self.parent[b]
        return b

    def make(self, key, parent):
        """the version of site `key`, below `parent` (a second visit with another parent moves it below their common ancestor)"""
        v = self.ids.get(key)
        if v is None:
            v = self.ids[key] = len(self.parent)
            self.parent[v] = parent
            self.changed = True
        elif self.parent[v] != parent:
            p = self.lca(self.parent[v], parent)
            if p == v or self.anc_or_eq(v, p):  # (degenerate: would make v its own ancestor)
                p = 0
            if p != self.parent[v]:
                self.parent[v] = p
                self.changed = True
        return v


def _exec_flow(ins):
    """Phase 1: (versions, version at every instruction's entry, version after it, successor lists, phi version per block entry)."""
    n = len(ins)
    addr_to_idx = {a: i for i, (a, _, _, _) in enumerate(ins)}
    succ = [[] for _ in range(n)]
    leaders = {0}
    for i, (a, size, op, ops) in enumerate(ins):
        if op in ("s_endpgm", "s_setpc_b64"):
            if i + 1 < n:
                leaders.add(i + 1)
            continue
        if op == "s_branch" or op.startswith("s_cbranch"):
            off = int(ops[-1])
            if off >= 0x8000:
                off -= 0x10000
            t = addr_to_idx.get(a + size + 4 * off)
            if t is not None:
                succ[i].append(t)
                leaders.add(t)
            if i + 1 < n:
                leaders.add(i + 1)
            if op == "s_branch":
                continue
        if i + 1 < n:
            succ[i].append(i + 1)
    order = sorted(leaders)
    block_of = {}
    blocks = []
    for k, st in enumerate(order):
        en = order[k + 1] if k + 1 < len(order) else n
        blocks.append((st, en))
        block_of[st] = k
    V = _ExecVersions()
    # state: (exec version, {sgpr index: (kind, version, half)}, {(vgpr, lane): content})
    state_in = [None] * len(blocks)
    state_in[0] = (0, {}, {})
    ex_in = [None] * n
    ex_out = [None] * n

    def pair(sg, regs):
        """content of a 64-bit scalar operand: (kind, version) when both halves hold the two halves of one tracked mask"""
        if len(regs) != 2:
            return None
        lo, hi = sg.get(regs[0]), sg.get(regs[1])
        if lo is None or hi is None or lo[:2] != hi[:2] or lo[2] != 0 or hi[2] != 1:
            return None
        return lo[:2]

    def set_pair(sg, regs, content):
        for h, r in enumerate(regs):
            if content is None:
                sg.pop(r, None)
            else:
                sg[r] = (content[0], content[1], h)

    def transfer(bi):
        st, en = blocks[bi]
        ex, sg, slots = state_in[bi]
        sg, slots = dict(sg), dict(slots)
        for i in range(st, en):
            _, _, op, ops = ins[i]
            ex_in[i] = ex
            new_ex = ex
            d0 = ops[0] if ops else ""
            if op.startswith("s_") and not op.startswith(_NO_SDST):
                src = ops[1:]
                dst_exec = d0 == "exec"
                uses_exec = "exec" in src
                content = None
                if op in ("s_and_saveexec_b64", "s_andn2_saveexec_b64", "s_xor_saveexec_b64", "s_or_saveexec_b64", "s_orn2_saveexec_b64",
                          "s_nand_saveexec_b64", "s_nor_saveexec_b64", "s_xnor_saveexec_b64", "s_andn1_saveexec_b64", "s_orn1_saveexec_b64"):
                    saved = ("eq", ex)
                    if op in ("s_and_saveexec_b64", "s_andn2_saveexec_b64", "s_andn1_saveexec_b64") and op != "s_andn1_saveexec_b64":
                        new_ex = V.make(("v", i), ex)
                    elif op == "s_or_saveexec_b64":
                        c = pair(sg, _sregs(src[0])) if src else None
                        new_ex = V.lca(ex, c[1]) if c else V.make(("t", i), 0)
                    else:
                        new_ex = V.make(("t", i), 0)
                    for r in _sregs(d0):
                        sg.pop(r, None)
                    set_pair(sg, _sregs(d0), saved)
                    ex_out[i] = ex = new_ex
                    continue
                if dst_exec:
                    other = [o for o in src if o != "exec"]
                    c = pair(sg, _sregs(other[0])) if other else None
                    if op in ("s_and_b64", "s_andn2_b64") and uses_exec and (op == "s_and_b64" or src[0] == "exec"):
                        new_ex = V.make(("v", i), ex)  # exec & x, exec & ~x: lanes only leave
                    elif op == "s_xor_b64" and uses_exec and c is not None and V.anc_or_eq(ex, c[1]):
                        new_ex = V.make(("v", i), ex)  # the `else` mask: exec ^ (lanes inside exec)
                    elif op == "s_or_b64" and uses_exec and c is not None:
                        new_ex = V.lca(ex, c[1])       # a restore: back to the version the register holds lanes of
                    elif op == "s_mov_b64" and c is not None:
                        new_ex = c[1] if c[0] == "eq" else V.make(("v", i), c[1])
                    else:
                        new_ex = V.make(("t", i), 0)
                    ex_out[i] = ex = new_ex
                    continue
                if d0 in ("exec_lo", "exec_hi"):
                    ex_out[i] = ex = V.make(("t", i), 0)
                    continue
                dregs = _sregs(d0)
                if len(dregs) == 2:
                    a = src[0] if src else ""
                    b = src[1] if len(src) > 1 else ""
                    ca = ("eq", ex) if a == "exec" else pair(sg, _sregs(a))
                    cb = ("eq", ex) if b == "exec" else pair(sg, _sregs(b))
                    if op == "s_mov_b64":
                        content = ("sub", ex) if a in ("0", "0x0") else ca  # (no lanes: inside any version; the seed of a loop's exit mask)
                    elif op in ("s_and_b64", "s_andn2_b64"):
                        # x & y lies inside both; x & ~y inside x
                        cands = [c for c in ((ca, cb) if op == "s_and_b64" else (ca,)) if c is not None]
                        if cands:
                            best = max(cands, key=lambda c: len(V.depth_path(c[1])))
                            content = ("sub", best[1])
                    elif op in ("s_or_b64", "s_xor_b64") and ca is not None and cb is not None:
                        content = ("sub", V.lca(ca[1], cb[1]))
                for r in dregs:
                    sg.pop(r, None)
                if content is not None:
                    set_pair(sg, dregs, content)
                if op in ("s_mov_b32",) and len(dregs) == 1 and len(src) == 1:
                    sr = _sregs(src[0])
                    if len(sr) == 1 and sr[0] in sg:
                        sg[dregs[0]] = sg[sr[0]]
            elif op == "v_writelane_b32":
                m = re.fullmatch(r"v(\d+)", d0)
                sr = _sregs(ops[1]) if len(ops) > 1 else ()
                lane = ops[2] if len(ops) > 2 else ""
                if m and lane.isdigit():
                    key = (int(m.group(1)), int(lane))
                    if len(sr) == 1 and sr[0] in sg:
                        slots[key] = sg[sr[0]]
                    else:
                        slots.pop(key, None)
                elif m:
                    for k in [k for k in slots if k[0] == int(m.group(1))]:
                        del slots[k]
            elif op == "v_readlane_b32":
                dregs = _sregs(d0)
                m = re.fullmatch(r"v(\d+)", ops[1]) if len(ops) > 1 else None
                lane = ops[2] if len(ops) > 2 else ""
                for r in dregs:
                    sg.pop(r, None)
                if m and lane.isdigit() and len(dregs) == 1 and (int(m.group(1)), int(lane)) in slots:
                    sg[dregs[0]] = slots[(int(m.group(1)), int(lane))]
            else:
                if op.startswith("v_cmpx"):
                    new_ex = V.make(("v", i), ex)
                # scalar destinations of vector instructions (compares, carries, readfirstlane): operand 0 or 1
                for o in ops[:2]:
                    for r in _sregs(o):
                        sg.pop(r, None)
                if op.startswith("v_cmp_") and len(_sregs(d0)) == 2:
                    set_pair(sg, _sregs(d0), ("sub", ex))  # a compare writes 0 for the lanes outside EXEC
                # a vector register that is written as a whole no longer holds spilled scalars
                if slots and not op.startswith(("buffer_store", "global_store", "flat_store", "ds_write", "scratch_store", "s_")):
                    for v in _vgprs(d0):
                        for k in [k for k in slots if k[0] == v]:
                            del slots[k]
                    if op == "v_swap_b32" and len(ops) > 1:
                        for v in _vgprs(ops[1]):
                            for k in [k for k in slots if k[0] == v]:
                                del slots[k]
            ex_out[i] = ex = new_ex
        return ex, sg, slots

    def merge(bi, incoming):
        ex, sg, slots = incoming
        cur = state_in[bi]
        if cur is None:
            state_in[bi] = (ex, sg, slots)
            return True
        cex, csg, cslots = cur
        nex = cex
        if ex != cex:
            phi = V.ids.get(("p", bi))
            if phi is not None and cex == phi:
                nex = V.make(("p", bi), V.lca(V.parent[phi] if V.parent[phi] is not None else 0, ex)) if not V.anc_or_eq(phi, ex) or ex != phi else phi
            else:
                nex = V.make(("p", bi), V.lca(cex, ex))
        nsg = {k: v for k, v in csg.items() if sg.get(k) == v}
        nslots = {k: v for k, v in cslots.items() if slots.get(k) == v}
        if nex != cex or len(nsg) != len(csg) or len(nslots) != len(cslots):
            state_in[bi] = (nex, nsg, nslots)
            return True
        return False

    for _ in range(50):  # (the version forest itself can move: repeat the flow until it does not)
        V.changed = False
        work = [0]
        queued = {0}
        while work:
            bi = work.pop()
            queued.discard(bi)
            out = transfer(bi)
            st, en = blocks[bi]
            for t in succ[en - 1]:
                tb = block_of[t]
                if merge(tb, out) and tb not in queued:
                    work.append(tb)
                    queued.add(tb)
        if not V.changed:
            break
    entry_version = [state_in[k][0] if state_in[k] is not None else None for k in range(len(blocks))]
    return V, ex_in, ex_out, succ, blocks, block_of, entry_version


_A_DEFS = ("v_accvgpr_write", "v_accvgpr_mov", "ds_read", "global_load", "buffer_load", "flat_load", "scratch_load", "v_mfma", "v_smfmac")


def parked_under_divergence(asm, kernels=None):
    """-> [(kernel, write, read)]: accumulator-register values that can be read under an EXEC wider than the one they were written under
    (see the comment above).  kernels: substrings of the (mangled) names to look at; None = every kernel that has a `v_accvgpr_write`."""
    bad = []

    def want(name, ops):
        if kernels is not None and not any(k in name for k in kernels):
            return False
        return any(o.startswith("v_accvgpr_write") for o in ops)

    for name, ins in _parse_kernels(asm, want):
        V, ex_in, ex_out, succ, blocks, block_of, entry_version = _exec_flow(ins)
        # phase 2: per accumulator register the version whose lanes are all known to hold a written value (`cov`: a MUST analysis over
        # the control-flow graph), and one site that wrote it (for the report).  A write under version W: W at or above cov -> cov = W (every
        # lane rewritten); below or beside it -> cov stays (a partial update of a value whose home is the accumulator register: the other
        # lanes keep what an earlier, wider write gave them).  An EXEC write or a join to a version that is not inside cov ends the dynamic
        # instance cov named: the value is no longer known to be whole (None).
        NONE = -1
        state_in = [None] * len(blocks)
        state_in[0] = {}
        reported = set()

        def through(cov, version):
            return {a: (w if w[0] != NONE and V.anc_or_eq(w[0], version) else (NONE, w[1])) for a, w in cov.items()}

        def meet(x, y):
            if x[0] == NONE or y[0] == NONE:
                return (NONE, x[1])
            if V.anc_or_eq(x[0], y[0]):
                return y
            if V.anc_or_eq(y[0], x[0]):
                return x
            return (NONE, x[1])

        def run(bi, report):
            st, en = blocks[bi]
            cov = dict(state_in[bi])
            for i in range(st, en):
                _, _, op, ops = ins[i]
                if ex_in[i] is None:
                    return None
                is_mfma = op.startswith(("v_mfma", "v_smfmac"))
                aops = [(k, _aregs(o)) for k, o in enumerate(ops) if o.startswith("a")]
                if aops:
                    is_def = op.startswith(_A_DEFS)
                    rv = 0 if is_mfma else ex_in[i]
                    if report:
                        for k, regs in aops:
                            if is_def and k == 0:
                                continue
                            for a in regs:
                                w = cov.get(a, (NONE, None))
                                if (w[0] == NONE or not V.anc_or_eq(w[0], rv)) and (a, i) not in reported:
                                    reported.add((a, i))
                                    d = ins[w[1]] if w[1] is not None else ("", 0, "(never written)", [])
                                    bad.append((name, d[2] + " " + ", ".join(d[3]), op + " " + ", ".join(ops)))
                    if is_def and aops[0][0] == 0:
                        wv = 0 if is_mfma else ex_in[i]
                        for a in aops[0][1]:
                            old = cov.get(a)
                            if old is None or old[0] == NONE or V.anc_or_eq(wv, old[0]):
                                cov[a] = (wv, i)
                if ex_out[i] != ex_in[i]:
                    cov = through(cov, ex_out[i])
            return cov

        work, queued = [0], {0}
        while work:
            bi = work.pop()
            queued.discard(bi)
            out = run(bi, False)
            if out is None:
                continue
            en = blocks[bi][1]
            for t in succ[en - 1]:
                tb = block_of[t]
                inc = through(out, entry_version[tb]) if entry_version[tb] != ex_out[en - 1] else out
                cur = state_in[tb]
                if cur is None:
                    state_in[tb] = dict(inc)
                    changed = True
                else:
                    changed = False
                    for a, w in inc.items():
                        # (a register no instruction has written on the other path constrains nothing: the compiler reads no register it
                        # never wrote, except along paths its own correlated branches rule out -- `if (unit) A; if (!unit) B; use`)
                        m = meet(cur[a], w) if a in cur else w
                        if cur.get(a) != m:
                            cur[a] = m
                            changed = True
                if changed and tb not in queued:
                    work.append(tb)
                    queued.add(tb)
        for bi in range(len(blocks)):
            if state_in[bi] is not None:
                run(bi, True)
    return bad


def m0_conflicts(asm):
    """The both-teams Q-network tick issues its LDS transfers from inline assembly (`s_mov_b32 m0, sN` + `global_load_lds_dwordx4`:
    susnet_qnet.h qnet_swap_issue) WITHOUT telling the compiler that M0 changes (it is a reserved register: not clobberable).  That is
    only sound while nothing else in the library reads or writes M0: -> every line that mentions m0 and is not that move."""
    bad = []
    for ln in asm.splitlines():
        code = ln.split("//")[0]
        if re.search(r"\bm0\b", code) and not re.search(r"\bs_mov_b32\s+m0,\s*s\d+", code):
            bad.append(code.strip())
    return bad


def scratch_instructions(asm):
    return [ln.strip() for ln in asm.splitlines() if re.search(r"\bscratch_", ln.split("//")[0])]


# The kernels bench.py's lines are made of (fused rollout, packed record).  VERDICT r02 item 1: at most 16 spilled SGPRs, no
# accumulation registers, no spilled VGPRs; and small enough for the instruction cache (64 KB per pair of CUs).
HEADLINE_KERNELS = ("k_rollout_duel<PhiloxRng, 6, false>", "k_rollout_duel<PhiloxRng, 4, false>", "k_rollout_duel<PhiloxRng, 6, true>", "k_rollout_swar<Spec<3, 4, 0, 1, -1, 1>, 4, PhiloxRng>",
                    "k_rollout_swar2<Spec<8, 4, 0, 1, -1, 2>, 4, PhiloxRng>", "k_rollout_swar<Spec<5, 5, 2, 1, -1, 1>, 4, PhiloxRng>")
HEADLINE_LIMITS = dict(sgpr_spill=16, agpr=0, vgpr_spill=0, v_accvgpr=0, scratch=0, code_bytes=48 * 1024)


def check_limits(rows):
    """-> list of violations of HEADLINE_LIMITS among HEADLINE_KERNELS (also: a headline kernel that is missing)."""
    by = {r["name"]: r for r in rows}
    bad = []
    for k in HEADLINE_KERNELS:
        if k not in by:
            bad.append(f"{k}: not in the library")
            continue
        for key, lim in HEADLINE_LIMITS.items():
            if by[k].get(key, 0) > lim:
                bad.append(f"{k}: {key} = {by[k].get(key)} > {lim}")
    # the LDS tables are addressed from 0 (lds_table_addr): no kernel that uses them may declare static LDS
    for r in rows:
        if r.get("lds_static", 0) != 0 and not r["name"].startswith("k_reduce_lifetime"):
            bad.append(f"{r['name']}: static LDS ({r['lds_static']} B) under the dynamic table area")
    return bad


def _analyse(obj: str):
    """One code object: (kernel rows, problems, wide stores seen) from ONE disassembly (the library holds some forty code objects:
    analysed in parallel -- the work is in the llvm tools' subprocesses)."""
    asm = disassemble(obj)
    problems = []
    hz = store_data_hazards(asm)
    if hz:
        problems.append(f"{os.path.basename(obj)}: {len(hz)} wide-store data hazards, e.g. {hz[:2]}")
    pk = parked_under_divergence(asm)
    if pk:
        problems.append(f"{os.path.basename(obj)}: {len(pk)} accumulator-register values read under a wider EXEC than they were written under, in "
                        f"{sorted(set(demangle([b[0] for b in pk]).values()))[:4]}, e.g. {[b[1:] for b in pk[:2]]}")
    m0 = m0_conflicts(asm)
    if m0:
        problems.append(f"{os.path.basename(obj)}: {len(m0)} uses of M0 beside the LDS-transfer moves (qnet_swap_issue writes it unannounced), e.g. {m0[:2]}")
    sc = scratch_instructions(asm)
    if sc:
        problems.append(f"{os.path.basename(obj)}: {len(sc)} scratch instructions, e.g. {sc[:2]}")
    notes = parse_notes(obj)
    dis = parse_disasm(obj, asm)
    names = demangle([k["mangled"] for k in notes])
    for k in notes:
        k["name"] = short_name(names.get(k["mangled"], k["mangled"]))
        k.update(dis.get(k["mangled"], {}))
        k["spill_moves"] = k.get("v_readlane", 0) + k.get("v_writelane", 0) + k.get("v_accvgpr", 0) + k.get("scratch", 0)
    return notes, problems, asm.count("_store_dwordx4")


def _analyse_all(lib: str):
    from concurrent.futures import ThreadPoolExecutor

    with tempfile.TemporaryDirectory() as wd:
        objs = code_objects(lib, wd)
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            parts = list(pool.map(_analyse, objs))
    rows = sorted((k for notes, _, _ in parts for k in notes), key=lambda r: r["name"])
    problems = [p for _, ps, _ in parts for p in ps]
    return objs, rows, problems, sum(n for _, _, n in parts)


def verify_library(lib: str = LIB):
    """Everything build_hip.build() requires of a freshly linked library; returns (rows, problems)."""
    objs, rows, problems, stores = _analyse_all(lib)
    if not objs:
        problems.append("no gfx950 code object in the library")
    elif stores < 100:
        problems.append("disassembly looks empty")
    problems += check_limits(rows)
    return rows, problems


def collect(lib: str = LIB):
    return _analyse_all(lib)[1]


COLS = ["vgpr", "agpr", "sgpr", "sgpr_spill", "vgpr_spill", "scratch_bytes", "instructions", "code_bytes", "v_readlane", "v_writelane", "v_accvgpr",
        "scratch", "s_nop", "s_waitcnt"]


def table(rows, md=False):
    if md:
        out = ["| kernel | " + " | ".join(COLS) + " |", "|---|" + "---|" * len(COLS)]
        for r in rows:
            out.append("| `" + r["name"] + "` | " + " | ".join(str(r.get(c, "")) for c in COLS) + " |")
        return "\n".join(out)
    w = max(len(r["name"]) for r in rows) if rows else 10
    out = [f"{'kernel':<{w}} " + " ".join(f"{c:>13}" for c in COLS)]
    for r in rows:
        out.append(f"{r['name']:<{w}} " + " ".join(f"{r.get(c, ''):>13}" for c in COLS))
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--match", action="append", default=[])
    ap.add_argument("--json")
    ap.add_argument("--md")
    args = ap.parse_args()
    if not tools_available():
        sys.exit("llvm-objdump / llvm-readelf (under " + LLVM + ") or c++filt not found")
    rows = collect(args.lib)
    if args.match:
        rows = [r for r in rows if all(m in r["name"] for m in args.match)]
    print(table(rows))
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)
    if args.md:
        open(args.md, "w").write(table(rows, md=True) + "\n")
    bad = check_limits(collect(args.lib))
    if bad:
        sys.exit("limits violated:\n  " + "\n  ".join(bad))


if __name__ == "__main__":
    main()
