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


EXEC_NARROWING = ("s_and_saveexec_b64", "s_andn2_saveexec_b64", "s_or_saveexec_b64", "s_xor_saveexec_b64")


def parked_under_divergence(asm, kernels=("k_qnet",)):
    """`v_accvgpr_write` that may execute under a narrowed EXEC, in the kernels that run at the register limit.

    There the register allocator parks long-lived values in accumulator registers around the matrix section.  A parking move it places inside a
    divergent region executes under that region's EXEC: the lanes outside keep whatever the accumulator held, and read it back as the value later
    (measured: the row index of k_qnet_step's Q-row store, parked inside `if (b + 32 < B)`; the last, partial wave of a batch stored through a wild
    pointer).  The Q-network code is written without divergent branches for that reason; this check keeps it so.
    Method: the kernel's control-flow graph from the branch offsets; forward data flow of "EXEC may be narrowed" (set by s_*_saveexec_b64 and by
    and / andn2 / xor into exec, cleared by `s_or_b64 exec, exec, ...` and `s_mov_b64 exec, ...`; union over predecessors).  An inner region's end
    clears the flag although an outer one may still be open: the check can miss, not false-alarm on straight code.
    Validated against: AMD clang 22.0.0git (ROCm 7.2.0)."""
    bad = []
    cur, ins = None, []

    def finish():
        if cur is None or not ins:
            return
        addr_to_idx = {a: i for i, (a, _, _) in enumerate(ins)}
        n = len(ins)
        succ = [[] for _ in range(n)]
        for i, (a, size, text) in enumerate(ins):
            op = text.split()[0]
            nxt = a + size
            if op in ("s_endpgm", "s_setpc_b64"):
                continue
            if op == "s_branch" or op.startswith("s_cbranch"):
                off = int(text.split()[-1])
                if off >= 0x8000:
                    off -= 0x10000
                t = addr_to_idx.get(nxt + 4 * off)
                if t is not None:
                    succ[i].append(t)
                if op == "s_branch":
                    continue
            if i + 1 < n:
                succ[i].append(i + 1)
        state_in = [None] * n  # None: not reached yet
        state_in[0] = False
        work = [0]
        while work:
            i = work.pop()
            st = state_in[i]
            text = ins[i][2]
            op = text.split()[0]
            ops = text.split(None, 1)[1].replace(" ", "") if " " in text else ""
            if op in EXEC_NARROWING or (op in ("s_and_b64", "s_andn2_b64", "s_xor_b64") and ops.startswith("exec,")):
                st = True
            elif op in ("s_or_b64", "s_mov_b64") and ops.startswith("exec,"):
                st = False
            for t in succ[i]:
                new = st if state_in[t] is None else (state_in[t] or st)
                if new != state_in[t]:
                    state_in[t] = new
                    work.append(t)
        for i, (_, _, text) in enumerate(ins):
            if state_in[i] and text.startswith("v_accvgpr_write_b32"):
                bad.append((cur, text))

    for ln in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", ln)
        if m:
            finish()
            cur, ins = (m.group(1) if any(k in m.group(1) for k in kernels) else None), []
            continue
        if cur is None or "\t" not in ln or "//" not in ln:
            continue
        text, tail = ln.split("//", 1)
        text = text.strip()
        mm = re.match(r"\s*([0-9A-Fa-f]+):\s*(.*)$", tail)
        if not text or not mm:
            continue
        ins.append((int(mm.group(1), 16), 4 * sum(1 for w in mm.group(2).split() if re.fullmatch(r"[0-9A-Fa-f]{8}", w)), text))  # (a branch line ends in "<symbol+0x..>")
    finish()
    return bad


def scratch_instructions(asm):
    return [ln.strip() for ln in asm.splitlines() if re.search(r"\bscratch_", ln.split("//")[0])]


# The kernels bench.py's lines are made of (fused rollout, packed record).  VERDICT r02 item 1: at most 16 spilled SGPRs, no
# accumulation registers, no spilled VGPRs; and small enough for the instruction cache (64 KB per pair of CUs).
HEADLINE_KERNELS = ("k_rollout_duel<PhiloxRng, 4>", "k_rollout_swar<Spec<3, 4, 0, 1, -1, 1>, 4, PhiloxRng>",
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
        problems.append(f"{os.path.basename(obj)}: {len(pk)} accumulator-register parking moves inside divergent regions, e.g. {pk[:2]}")
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
