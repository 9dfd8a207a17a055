#!/usr/bin/env python3
"""Register / spill / code-size table of every kernel in the shipped gfx950 code objects.

    python tools/kernel_resources.py [--match SUBSTR] [--json OUT] [--md OUT]

Reads the ELF notes (`llvm-readelf --notes`: vgpr / agpr / sgpr counts, spill counts, private segment, LDS) and the
disassembly (`llvm-objdump -d`: static instruction count and the moves register pressure costs -- `v_readlane` /
`v_writelane` = SGPR spills to vector lanes, `v_accvgpr_*` = VGPR spills to accumulation registers, `scratch_*` = memory)
of the gfx950 code objects embedded in sus-net_amd/libsusnet_hip.so.  Used by build_hip.py (hazard scan), by
tests/test_capi_abi.py (the limits the headline kernels must keep) and to write profiles/rNN_kernel_resources.md.
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "sus-net_amd", "libsusnet_hip.so")


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


def parse_disasm(obj: str):
    """mangled kernel name -> static instruction statistics"""
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", obj], capture_output=True, text=True, check=True).stdout
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
            enc = body[1].split(":")[-1].split()
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


def collect(lib: str = LIB):
    rows = []
    with tempfile.TemporaryDirectory() as wd:
        for obj in code_objects(lib, wd):
            notes = parse_notes(obj)
            dis = parse_disasm(obj)
            names = demangle([k["mangled"] for k in notes])
            for k in notes:
                k["name"] = short_name(names.get(k["mangled"], k["mangled"]))
                k.update(dis.get(k["mangled"], {}))
                k["spill_moves"] = k.get("v_readlane", 0) + k.get("v_writelane", 0) + k.get("v_accvgpr", 0) + k.get("scratch", 0)
                rows.append(k)
    rows.sort(key=lambda r: r["name"])
    return rows


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


if __name__ == "__main__":
    main()
