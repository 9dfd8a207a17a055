#!/usr/bin/env python3
"""Where do a kernel's instructions come from?  Static attribution of gfx950 instructions to source lines.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -gline-tables-only -S --cuda-device-only sus-net_amd/csrc/inst_cfg3.hip -o /tmp/cfg3.s
    python tools/isa_sections.py /tmp/cfg3.s --kernel 'k_rollout_swar.*Li4ENS_9PhiloxRng' [--loop] [--by-line]

Reads the `.loc` directives of the assembly (with their inlined-at chains), counts instructions per (file, line) and per class
(VALU / SALU / LDS / VMEM / branch), and sums them over named source ranges (SECTIONS below: the sections of the byte-parallel step).
--loop restricts the count to the kernel's largest loop (from the target label of the last backward branch to that branch): the
tick loop of a fused rollout.  Static counts: a branch not taken at run time still counts.
"""
from __future__ import annotations

import argparse
import collections
import re
import sys

# (file suffix, first line, last line, name) -- kept next to the sources they describe; a line outside every range is listed by file
SECTIONS: list[tuple[str, int, int, str]] = []


def classify(op: str) -> str:
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "BRANCH"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    return "OTHER"


def parse(path: str, kernel_re: str):
    files: dict[int, str] = {}
    rx_file = re.compile(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?')
    rx_loc = re.compile(r"^\s*\.loc\s+(\d+)\s+(\d+)")
    rx_frame = re.compile(r"([^\s\[\]@;]+):(\d+):\d+")
    rx_lab = re.compile(r"^([.\w$]+):")
    rx_ins = re.compile(r"^\s+([a-z_0-9]+)\b(.*)$")
    krx = re.compile(kernel_re)
    inside = False
    cur = (0, 0, ())
    insts = []  # (index, op, args, file, line, label-before)
    labels = {}
    with open(path) as f:
        for ln in f:
            m = rx_file.match(ln)
            if m:
                files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
                continue
            m = rx_lab.match(ln)
            if m and not ln.startswith("\t"):
                name = m.group(1)
                if not inside and krx.search(name) and not name.startswith("."):
                    inside = True
                    continue
                if inside:
                    if name.startswith(".Lfunc_end"):
                        break
                    labels[name] = len(insts)
                continue
            if not inside:
                continue
            m = rx_loc.match(ln)
            if m:
                # the whole inlined-at chain, outermost frame first: [(file, line), ...]
                chain = [(fn.split("/")[-1], int(l)) for fn, l in rx_frame.findall(ln.split(";", 1)[1])] if ";" in ln else []
                cur = (int(m.group(1)), int(m.group(2)), tuple(reversed(chain)))
                continue
            if ln.lstrip().startswith((".", ";")):
                continue
            m = rx_ins.match(ln)
            if m:
                insts.append((m.group(1), m.group(2).split(";")[0].strip(), files.get(cur[0], "?"), cur[1], cur[2]))
    return insts, labels


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", required=True, help="regex on the mangled kernel symbol")
    ap.add_argument("--loop", action="store_true", help="only the largest loop of the kernel")
    ap.add_argument("--by-line", action="store_true")
    ap.add_argument("--range", action="append", default=[], help="FILE:LO-HI=NAME (repeatable): sum over a source range")
    args = ap.parse_args()
    insts, labels = parse(args.asm, args.kernel)
    if not insts:
        print("kernel not found", file=sys.stderr)
        return 1
    lo, hi = 0, len(insts)
    if args.loop:
        best = (0, 0, 0)
        for i, (op, a, *_r) in enumerate(insts):
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = a.split()[-1] if a else ""
                if tgt in labels and labels[tgt] <= i and i - labels[tgt] > best[0]:
                    best = (i - labels[tgt], labels[tgt], i + 1)
        lo, hi = best[1], best[2]
    body = insts[lo:hi]
    tot = collections.Counter(classify(op) for op, *_ in body)
    print(f"{len(body)} instructions ({'loop' if args.loop else 'kernel'}): " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())))
    ranges = []
    for r in args.range:
        m = re.match(r"([^:]+):(\d+)-(\d+)=(.*)", r)
        ranges.append((m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)))
    by_sec = collections.defaultdict(collections.Counter)
    by_line = collections.defaultdict(collections.Counter)
    for op, a, fn, line, chain in body:
        name = None
        # the first declared range that ANY frame of the inlined-at chain falls in (ranges in the order given: list the rare
        # branches -- reset, refill -- before the sections of the step they call into)
        for suf, l0, l1, nm in ranges:
            if any(f.endswith(suf) and l0 <= l <= l1 for f, l in (chain or ((fn, line),))):
                name = nm
                break
        if name is None:
            name = fn
        by_sec[name][classify(op)] += 1
        by_line[(fn, line)][classify(op)] += 1
    print(f"{'section':40s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'BR':>4s} {'WAIT':>5s}")
    for nm, cnt in sorted(by_sec.items(), key=lambda kv: -kv[1]["VALU"]):
        print(f"{nm:40s} {cnt['VALU']:6d} {cnt['SALU']:6d} {cnt['LDS']:5d} {cnt['VMEM']:5d} {cnt['BRANCH']:4d} {cnt['WAIT']:5d}")
    if args.by_line:
        print()
        for (fn, line), cnt in sorted(by_line.items()):
            print(f"{fn}:{line:<6d} " + " ".join(f"{k}={v}" for k, v in sorted(cnt.items())))
    return 0


if __name__ == "__main__":
    sys.exit(main())
