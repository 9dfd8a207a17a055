"""Condense the --pmc passes of tools/profile_cfg5.sh (gpurun_out/prof_<tag>/cfg5/pmc_*) into profiles/<tag>_qnet_counters.json: the network kernel's
counters per launch and per wave (what bench.py attaches to the cfg5 line as roofline.traffic).      python tools/summarize_qnet.py <tag>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "cfg5")
vals = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(base, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:  # (newest: gpurun merges)
        for r in csv.DictReader(open(f)):
            if "k_qnet<" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for k, v in vals.items()}
m = lambda k: c[k]["mean_per_launch"]
waves = m("SQ_WAVES")
per = {"mfma_f32_instructions": m("SQ_INSTS_VALU_MFMA_F32") / waves, "wave_cycles": 4.0 * m("SQ_WAVE_CYCLES") / waves,
       "mfma_busy_cycles": m("SQ_VALU_MFMA_BUSY_CYCLES") / waves, "valu": m("SQ_INSTS_VALU") / waves, "lds": m("SQ_INSTS_LDS") / waves}
per["mfma_busy_share"] = per["mfma_busy_cycles"] / per["wave_cycles"]
per["valu_without_mfma"] = per["valu"] - per["mfma_f32_instructions"]
out = {"kernel": "k_qnet<FlatRow<2,3,14>> (tools/qnet_bench.py, 65 536 envs)", "counters": c, "per_wave": per,
       "traffic_bytes_per_launch": (m("WRITE_SIZE") + 2.0 * m("FETCH_SIZE")) * 1024.0,
       "note": "SQ_WAVE_CYCLES ticks in quad-cycles (x 4), SQ_VALU_MFMA_BUSY_CYCLES in cycles (1 376 x 64); SQ_INSTS_VALU counts the MFMAs too; "
               "traffic = WRITE_SIZE + 2 x FETCH_SIZE (KiB units, FETCH doubled: MI355X_MICROARCH.md's gfx950 correction)"}
dst = os.path.join(ROOT, "profiles", f"{tag}_qnet_counters.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, json.dumps(per), out["traffic_bytes_per_launch"])
