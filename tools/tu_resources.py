#!/usr/bin/env python3
"""Register / spill figures of ONE translation unit without linking the library: compiles sus-net_amd/csrc/<unit>.hip with
-Rpass-analysis=kernel-resource-usage and prints one line per kernel (the loop for register-pressure work: ~25 s per unit).

    python tools/tu_resources.py inst_fam_a8_tag_ni2 [--match SUBSTR] [-D MACRO ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sus-net_amd"))
import isa_checks as ic  # noqa: E402

args = sys.argv[1:]
unit = args.pop(0)
match, defs = [], []
while args:
    a = args.pop(0)
    if a == "--match":
        match.append(args.pop(0))
    elif a == "-D":
        defs.append("-D" + args.pop(0))
src = os.path.join(ROOT, "sus-net_amd", "csrc", unit + ("" if unit.endswith(".hip") else ".hip"))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage",
       *defs, "-c", "-o", "/tmp/tu_resources.o", src]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in err.splitlines():
    m = re.search(r"remark: +(Function Name|[A-Za-z ]+\[?[A-Za-z]*\]?): +(\S+)", ln)
    if not m:
        if "error" in ln:
            print(ln)
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"mangled": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = ic.demangle([r["mangled"] for r in rows])
print(f"{'kernel':<78} {'VGPR':>5} {'AGPR':>5} {'SGPRsp':>6} {'VGPRsp':>6} {'scratch':>7}")
for r in rows:
    n = ic.short_name(names[r["mangled"]])
    if match and not all(m in n for m in match):
        continue
    print(f"{n[:78]:<78} {r.get('VGPRs', ''):>5} {r.get('AGPRs', ''):>5} {r.get('SGPRs Spill', ''):>6} {r.get('VGPRs Spill', ''):>6} {r.get('ScratchSize [bytes/lane]', ''):>7}")
