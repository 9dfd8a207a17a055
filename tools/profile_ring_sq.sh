#!/bin/bash
# Runs on the GPU box (via gpurun): SQ instruction / cycle counters of the replay-ring append kernels, row kernel and tile kernels side by
# side (tools/ring_bench.py creates one handle per SUSNET_RING_TILE value).   usage: PROF_TAG=r04ring tools/profile_ring_sq.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_${PROF_TAG:-cur}/ring_sq
rm -rf "$OUT" && mkdir -p "$OUT"
export RING_BENCH_VARIANTS=${RING_BENCH_VARIANTS:-default,groups=1,groups=4}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/ring_bench.py > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -- python3 tools/ring_bench.py > $OUT/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 tools/ring_bench.py > $OUT/pmc_sq2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE FETCH_SIZE --output-format csv -d $OUT/pmc_mem -- python3 tools/ring_bench.py > $OUT/pmc_mem.log 2>&1 || exit 1
rocprofv3 -L > $OUT/counters_available.txt 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
# kernel-trace order: per shape, per tile, 3 + 20 append launches (+ populate_fused's)
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for grp in ("pmc_sq1", "pmc_sq2", "pmc_mem"):
    for f in glob.glob(f"{out}/{grp}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_ring_append" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r.get("LDS_Block_Size", 0) or 0))
                rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/stats/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_ring_append" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r.get("LDS_Block_Size", 0) or 0))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key in sorted(rows):
    m = {k: sum(v) / len(v) for k, v in rows[key].items()}
    w = m.get("SQ_WAVES", 1)
    d = sorted(dur.get(key, [0]))
    print(key, f"launches={len(rows[key].get('SQ_WAVES', []))} median_us={d[len(d) // 2]:.1f}",
          " ".join(f"{k[3:]}={m[k] / w:.0f}" for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD") if k in m),
          "cycles/wave=%.0f" % (m.get("SQ_WAVE_CYCLES", 0) * 4 / w),
          " ".join(f"{k[3:]}={m[k] * 4 / w:.0f}" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS") if k in m),
          " ".join(f"{k}={m[k]:.0f}" for k in ("WRITE_SIZE", "FETCH_SIZE") if k in m))
PY
echo ring-sq-ok
