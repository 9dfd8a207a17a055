"""Condense gpurun_out/prof/* (rocprofv3 CSV) into small tracked files under profiles/.
usage: python tools/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(f"{src}/stats/*/*kernel_stats.csv")
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
summary = {}
for d in sorted(glob.glob(f"{src}/pmc_*")):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:80]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, cs in agg.items():
        if "rollout" not in k and "k_step" not in k and "k_sample" not in k:
            continue
        out[k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in cs.items()}
        out[k]["avg_duration_us"] = sum(dur[k]) / len(dur[k])
    summary[name] = out
json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
