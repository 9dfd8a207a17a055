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

# bench.py reads `roofline.traffic` from the summary that was committed BEFORE the profiled run; the bench lines kept
# next to this summary get the traffic of the SAME run (WRITE_SIZE + 2 x FETCH_SIZE, KiB -> bytes)
try:
    ksel = lambda grp, ctr: next(v[ctr]["mean_per_launch"] for k, v in summary[grp].items() if k.startswith("void k_rollout") and ", 2>" in k)
    traffic = (ksel("pmc_WRITE_SIZE", "WRITE_SIZE") + 2.0 * ksel("pmc_FETCH_SIZE", "FETCH_SIZE")) * 1024.0
    for name in (f"profiles/{tag}_bench_under_rocprofv3.json", f"profiles/{tag}_bench_unprofiled.json"):
        if os.path.exists(name):
            line = json.load(open(name))
            line["roofline"]["traffic"] = traffic
            line["roofline"]["traffic_source"] = f"{tag}_pmc_summary.json"
            json.dump(line, open(name, "w"))
    print("traffic per launch:", traffic)
except (KeyError, StopIteration) as exc:
    print("traffic not refreshed:", exc)
