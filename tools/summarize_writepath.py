"""Condense gpurun_out/prof_<tag>/writepath_<cfg>/<group>/ (tools/profile_writepath.sh: one rocprofv3 --pmc pass per counter
group) into profiles/<tag>_writepath_<cfg>.json: per-launch means of every counter for the fused rollout kernel, plus the
figures derived from them (requests per wave-tick, bytes per request, stall shares).
usage: python tools/summarize_writepath.py r03m cfg2 [record_bytes] [ticks]"""
import collections
import csv
import glob
import json
import sys

tag, cfg = sys.argv[1], sys.argv[2]
rec = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ticks = int(sys.argv[4]) if len(sys.argv) > 4 else 512
out = {"config": cfg, "record_bytes": rec, "ticks_per_launch": ticks, "groups": {}}
flat = {}
for d in sorted(glob.glob(f"gpurun_out/prof_{tag}/writepath_{cfg}/*/")):
    grp = d.rstrip("/").split("/")[-1]
    agg, dur = collections.defaultdict(list), []
    for f in glob.glob(d + "*/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "k_rollout" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if not agg:
        continue
    g = {c: sum(v) / len(v) for c, v in agg.items()}
    g["launches"] = len(dur)
    g["avg_duration_us"] = sum(dur) / len(dur)
    out["groups"][grp] = g
    flat.update({c: v for c, v in g.items() if c not in ("launches", "avg_duration_us")})
waves = flat.get("SQ_WAVES", 1024.0)
wt = waves * ticks
lanes = waves * 64
d = {}
if "TCP_TCC_WRITE_REQ" in flat:
    d["tcp_to_l2_write_requests_per_wave_tick"] = flat["TCP_TCC_WRITE_REQ"] / wt
    d["bytes_per_tcp_write_request"] = lanes * ticks * rec / flat["TCP_TCC_WRITE_REQ"]
if "SQ_INSTS_VMEM_WR" in flat:
    d["store_instructions_per_wave_tick"] = flat["SQ_INSTS_VMEM_WR"] / wt
if "SQ_WAIT_ANY" in flat and "SQ_WAVE_CYCLES" in flat:
    d["share_wave_cycles_waiting"] = flat["SQ_WAIT_ANY"] / flat["SQ_WAVE_CYCLES"]
if "TCC_EA0_WRREQ" in flat:
    d["l2_to_memory_write_requests_per_wave_tick"] = flat["TCC_EA0_WRREQ"] / wt
    d["share_64B_of_l2_to_memory_write_requests"] = flat.get("TCC_EA0_WRREQ_64B", 0.0) / flat["TCC_EA0_WRREQ"]
    d["bytes_per_l2_to_memory_write_request"] = lanes * ticks * rec / flat["TCC_EA0_WRREQ"]
if "TCC_WRITE" in flat and "TCC_REQ" in flat:
    d["l2_write_requests_per_wave_tick"] = flat["TCC_WRITE"] / wt
    d["share_writes_of_l2_requests"] = flat["TCC_WRITE"] / flat["TCC_REQ"]
out["derived"] = d
json.dump(out, open(f"profiles/{tag}_writepath_{cfg}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
