"""Condense tools/profile_aux.sh output (gpurun_out/prof_<tag>/{featurize,ring}/) into profiles/<tag>_aux_summary.json.

Per kernel of interest and per workload of the bench script (launches grouped by the bytes they wrote) the average duration from
the --kernel-trace pass, the WRITE_SIZE of the --pmc pass (KiB per dispatch, MI355X_MICROARCH.md's unit) and bytes written / time.
    python tools/summarize_aux.py <tag>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_ring_append", "k_ring_window", "k_featurize", "k_scent", "k_observe")


def short(name: str) -> str:
    name = name.replace("void ", "").replace("susnet::", "")
    return name.split("(")[0]


def main():
    tag = sys.argv[1]
    out = {}
    for work in ("featurize", "ring"):
        base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", work)
        # (gpurun MERGES a call's files into the local gpurun_out/: earlier sessions' files of the same pass may still lie there -- newest only)
        traces = sorted(glob.glob(os.path.join(base, "stats", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]
        pmcs = sorted(glob.glob(os.path.join(base, "pmc_WRITE_SIZE", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
        if not traces:
            continue
        # the i-th dispatch of a kernel is the same launch in both runs (same script, same order): pair them, then group a kernel's
        # launches by what they wrote -- one group per workload of the bench script
        dur, wr = defaultdict(list), defaultdict(list)
        for row in csv.DictReader(open(traces[0])):
            k = short(row["Kernel_Name"])
            if k.startswith(KERNELS):
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        for row in (csv.DictReader(open(pmcs[0])) if pmcs else []):
            k = short(row["Kernel_Name"])
            if k.startswith(KERNELS) and row["Counter_Name"] == "WRITE_SIZE":
                wr[k].append(float(row["Counter_Value"]))
        for k in sorted(dur):
            groups = defaultdict(list)
            paired = len(wr.get(k, [])) == len(dur[k])
            for i, us in enumerate(dur[k]):
                groups[round(wr[k][i] / 1024.0) if paired else -1].append(us)
            for mib in sorted(groups):
                us = sum(groups[mib]) / len(groups[mib])
                entry = {"launches": len(groups[mib]), "avg_us": us}
                if mib >= 0:
                    entry.update(written_MiB=mib, written_TBs=mib * 1048576.0 / us / 1e6)
                out.setdefault(k, []).append(entry)
        log = os.path.join(base, "unprofiled.log")
        if os.path.exists(log):
            out[f"{work}_bench_lines"] = [json.loads(l) for l in open(log) if l.startswith("{")]
    dst = os.path.join(ROOT, "profiles", f"{tag}_aux_summary.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(dst)
    for k, v in out.items():
        if not k.endswith("_lines"):
            for e in v:
                print(k, e)


if __name__ == "__main__":
    main()
