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
    stats = sorted(glob.glob(f"{cdir}/stats/*/*kernel_stats.csv"), key=os.path.getmtime)[-1:]  # (newest: see the counter files below)
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
        # (gpurun MERGES a call's files into the local gpurun_out/: an earlier session's files of the same pass may still lie there -- newest only)
        for f in sorted(glob.glob(f"{d}/*/*counter_collection.csv"), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0][:80]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        out = {}
        for k, cs in agg.items():
            if not any(x in k for x in ("rollout", "k_step", "k_sample", "k_featurize", "k_ring", "k_scent", "k_observe")):
                continue
            out[k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v), "sum": sum(v)} for c, v in cs.items()}
            # (every counter row of a dispatch repeats its timestamps: average over dispatches, not rows)
            out[k]["avg_duration_us"] = sum(dur[k]) / len(dur[k])
            # bench steps in the pass = launches that run (nearly) a whole step: the settle loop makes their number vary from run to run, and
            # under counter collection a long step can arrive as two dispatches (tag5: 482 + 30 ticks) -- the short ones are parts, not steps
            n_rows = len(next(iter(cs.values())))
            dd = dur[k][::max(1, len(dur[k]) // n_rows)][:n_rows]
            out[k]["bench_steps"] = sum(1 for x in dd if x >= 0.5 * max(dd))
            out[k]["launches_per_bench_step"] = n_rows / out[k]["bench_steps"]
        per[os.path.basename(d)] = out
    # derived per wave-tick figures of the fused rollout kernel (SQ_* cycle counters tick in quad-cycles)
    try:
        roll = lambda grp: next(v for k, v in per[grp].items() if "k_rollout" in k)
        s1, s2 = roll("pmc_sq1"), roll("pmc_sq2")
        waves = s1["SQ_WAVES"]["mean_per_launch"]
        # every bench step is 512 ticks; totals over the pass divided by the steps in it (see bench_steps above)
        wt = waves * 512 * s1["bench_steps"]
        wc, wc2 = s1["SQ_WAVE_CYCLES"]["sum"], None
        per["derived_per_wave_tick"] = {
            "waves": waves, "bench_steps_in_pass": s1["bench_steps"], "launches_per_bench_step": s1["launches_per_bench_step"],
            "valu": s1["SQ_INSTS_VALU"]["sum"] / wt, "salu": s1["SQ_INSTS_SALU"]["sum"] / wt,
            "lds": s1["SQ_INSTS_LDS"]["sum"] / wt, "vmem_wr": s1["SQ_INSTS_VMEM_WR"]["sum"] / wt,
            "branch": s1["SQ_INSTS_BRANCH"]["sum"] / wt, "cycles": 4.0 * wc / wt,
            # (second pass: shares of ITS wave cycles ~ the first pass's scaled by the steps of each)
            "share_active_inst_any": s2["SQ_ACTIVE_INST_ANY"]["sum"] / s2["bench_steps"] / (wc / s1["bench_steps"]),
            "share_wait_any": s2["SQ_WAIT_ANY"]["sum"] / s2["bench_steps"] / (wc / s1["bench_steps"]),
            "share_wait_inst_any": s2["SQ_WAIT_INST_ANY"]["sum"] / s2["bench_steps"] / (wc / s1["bench_steps"]),
        }
        rw, rf = roll("pmc_WRITE_SIZE"), roll("pmc_FETCH_SIZE")
        per["traffic_bytes_per_bench_step"] = (rw["WRITE_SIZE"]["sum"] / rw["bench_steps"] + 2.0 * rf["FETCH_SIZE"]["sum"] / rf["bench_steps"]) * 1024.0
        per["traffic_bytes_per_launch"] = per["traffic_bytes_per_bench_step"] / rw["launches_per_bench_step"]
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
