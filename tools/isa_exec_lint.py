#!/usr/bin/env python3
"""Lists accumulator-register parking moves (v_accvgpr_write / v_accvgpr_read, scratch stores / loads) that sit where EXEC may be narrowed.

Background: a kernel at the register limit (k_qnet_step: 256 + 244 registers) has its long-lived values parked in AGPRs by the register
allocator.  A parking move placed inside a divergent region runs under that region's EXEC; a value that is live in lanes outside the
region is lost there (seen once: the row index of the Q-row store, parked inside `if (b + 32 < B)`).  This scan is linear and
conservative (it does not follow branches): between an instruction that narrows EXEC and the next `s_or_b64 exec, exec, ...` every
parking move is reported with its position relative to the matrix section.

    python tools/isa_exec_lint.py [--lib sus-net_amd/libsusnet_hip.so] [--match k_qnet_step]
"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sus-net_amd"))
import isa_checks  # noqa: E402

NARROW = ("s_and_saveexec_b64", "s_andn2_saveexec_b64", "s_or_saveexec_b64", "s_xor_saveexec_b64")


def scan(body):
    narrowed, since = False, -1
    out = []
    mfma = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
    first, last = (mfma[0], mfma[-1]) if mfma else (-1, -1)
    for i, l in enumerate(body):
        op = l.split(None, 1)[0] if l else ""
        if op in NARROW or (op in ("s_and_b64", "s_andn2_b64", "s_xor_b64", "s_mov_b64") and l.split(None, 1)[1].startswith("exec")):
            narrowed, since = True, i
        elif op == "s_or_b64" and l.split(None, 1)[1].startswith("exec"):
            narrowed = False
        elif narrowed and (op in ("v_accvgpr_write_b32", "v_accvgpr_read_b32") or op.startswith("scratch_")):
            where = "before" if i < first else "inside" if i <= last else "after"
            out.append((i, since, where, l))
    return out, first, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "sus-net_amd", "libsusnet_hip.so"))
    ap.add_argument("--match", default="k_qnet")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    bad = 0
    with tempfile.TemporaryDirectory() as wd:
        for obj in isa_checks.code_objects(args.lib, wd):
            txt = subprocess.check_output([isa_checks.OBJDUMP if hasattr(isa_checks, "OBJDUMP") else "/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", obj]).decode()
            if args.match not in txt:
                continue
            lines = txt.split("\n")
            heads = [i for i, l in enumerate(lines) if l.endswith(">:")]
            for n, st in enumerate(heads):
                name = lines[st].split("<", 1)[1][:-2]
                if args.match not in name:
                    continue
                end = heads[n + 1] if n + 1 < len(heads) else len(lines)
                body = [l.split("//")[0].strip() for l in lines[st + 1:end]]
                hits, first, last = scan(body)
                try:
                    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
                except OSError:
                    demangled = name
                per = {"before": 0, "inside": 0, "after": 0}
                for h in hits:
                    per[h[2]] += 1
                print(f"{demangled[:110]}: matrix section at {first}..{last} of {len(body)}; parking moves under narrowed EXEC: {per}")
                bad += per["before"] + per["inside"]
                if args.verbose:
                    for i, since, where, l in hits:
                        print(f"    {i:6d} (EXEC narrowed at {since}) [{where} the matrix section] {l}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
