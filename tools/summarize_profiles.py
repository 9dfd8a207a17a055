"""Condense gpurun_out/prof/<config>/* (rocprofv3 CSV, tools/profile_gpu.sh) into small tracked files under profiles/.
usage: python tools/summarize_profiles.py r02 [gpurun_out/prof_r02]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
summary = {}
for cdir in sorted(glob.glob(f"{root}/*")):
    if not os.path.isdir(cdir):
        continue
    cfg = os.path.basename(cdir)
    stats = glob.glob(f"{cdir}/stats/*/*kernel_stats.csv")
    if stats:
        shutil.copy(stats[0], f"profiles/{tag}_{cfg}_kernel_stats.csv")
    if not cfg.startswith(("cfg", "tag")):  # (featurize / ring / writepath_*: tools/profile_aux.sh, tools/profile_writepath.sh: kernel stats only here)
        continue
    # the JSON line bench.py printed inside the profiled --stats run
    log = f"{cdir}/stats.log"
    per = {}
    if os.path.exists(log):
        for ln in open(log, errors="replace"):
            if ln.startswith('{"metric"'):
                open(f"profiles/{tag}_{cfg}_bench_under_rocprofv3.json", "w").write(ln)
                c = json.loads(ln)["config"]
                # what was profiled: bench.py attaches these PMC bytes only to a run of the same layout / batch / ticks
                per["command"] = {"packed": c.get("trajectory_layout", "").startswith(("packed", "compact")), "batch": c.get("batch_per_gpu"),
                                  "ticks": c.get("ticks_per_launch"), "bytes_per_env_step": json.loads(ln)["roofline"].get("bytes_per_env_step"),
                                  "trajectory_layout": c.get("trajectory_layout")}
    for d in sorted(glob.glob(f"{cdir}/pmc_*")):
        if not os.path.isdir(d):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for f in glob.glob(f"{d}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0][:80]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        out = {}
        for k, cs in agg.items():
            if not any(x in k for x in ("rollout", "k_step", "k_sample", "k_featurize", "k_ring", "k_scent", "k_observe")):
                continue
            out[k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in cs.items()}
            # (every counter row of a dispatch repeats its timestamps: average over dispatches, not rows)
            out[k]["avg_duration_us"] = sum(dur[k]) / len(dur[k])
            # (the profiled command runs 33 bench steps; a step split into consecutive launches shows as a multiple of 33)
            n = len(dur[k])
            out[k]["launches_per_bench_step"] = n // 33 if n % 33 == 0 and n >= 33 else 1
        per[os.path.basename(d)] = out
    # derived per wave-tick figures of the fused rollout kernel (SQ_* cycle counters tick in quad-cycles)
    try:
        roll = lambda grp: next(v for k, v in per[grp].items() if "k_rollout" in k)
        s1, s2 = roll("pmc_sq1"), roll("pmc_sq2")
        waves = s1["SQ_WAVES"]["mean_per_launch"]
        # the profiled command runs 33 bench steps of 512 ticks (5 warm-up + 20 timed + 8 event-pair launches); a step whose
        # record array would pass 2 GiB is split into consecutive launches (tag5: two), so a launch holds fewer ticks
        launches = s1["SQ_WAVES"]["launches"]
        ticks = 512 * 33 / launches if launches % 33 == 0 else 512
        wt = waves * ticks
        wc = s1["SQ_WAVE_CYCLES"]["mean_per_launch"]
        per["derived_per_wave_tick"] = {
            "waves": waves, "ticks_per_launch": ticks,
            "valu": s1["SQ_INSTS_VALU"]["mean_per_launch"] / wt, "salu": s1["SQ_INSTS_SALU"]["mean_per_launch"] / wt,
            "lds": s1["SQ_INSTS_LDS"]["mean_per_launch"] / wt, "vmem_wr": s1["SQ_INSTS_VMEM_WR"]["mean_per_launch"] / wt,
            "branch": s1["SQ_INSTS_BRANCH"]["mean_per_launch"] / wt, "cycles": 4.0 * wc / wt,
            "share_active_inst_any": s2["SQ_ACTIVE_INST_ANY"]["mean_per_launch"] / wc,
            "share_wait_any": s2["SQ_WAIT_ANY"]["mean_per_launch"] / wc,
            "share_wait_inst_any": s2["SQ_WAIT_INST_ANY"]["mean_per_launch"] / wc,
        }
        w = roll("pmc_WRITE_SIZE")["WRITE_SIZE"]["mean_per_launch"]
        f = roll("pmc_FETCH_SIZE")["FETCH_SIZE"]["mean_per_launch"]
        per["traffic_bytes_per_launch"] = (w + 2.0 * f) * 1024.0
        # (a bench step whose record array would pass 2 GiB runs as consecutive launches: tag5 takes two)
        per["traffic_bytes_per_bench_step"] = per["traffic_bytes_per_launch"] * roll("pmc_WRITE_SIZE").get("launches_per_bench_step", 1)
    except (KeyError, StopIteration, ZeroDivisionError) as exc:
        per["derived_error"] = repr(exc)
    summary[cfg] = per
json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
for cfg, per in summary.items():
    print(cfg, json.dumps(per.get("derived_per_wave_tick", per.get("derived_error"))), per.get("traffic_bytes_per_launch"))
    # bench lines kept next to this summary get the traffic of the SAME profiling session
    name = f"profiles/{tag}_{cfg}_bench_under_rocprofv3.json"
    if os.path.exists(name) and per.get("traffic_bytes_per_bench_step"):
        line = json.load(open(name))
        t = per["traffic_bytes_per_bench_step"]
        line["roofline"]["traffic"] = t
        line["roofline"]["traffic_source"] = f"{tag}_pmc_summary.json"
        if line["roofline"].get("avg_launch_us"):
            line["roofline"]["traffic_frac"] = t / (line["roofline"]["avg_launch_us"] * 1e-6) / 8e12
        json.dump(line, open(name, "w"))
