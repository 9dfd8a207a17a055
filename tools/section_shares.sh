#!/bin/bash
# Runs on the GPU box (via gpurun): what each section of the byte-parallel step costs, from DIAGNOSTIC builds that leave one section out
# (tools/build_variant.sh lib_skip_X -DSUSNET_EXP_SKIP_X for X in KILL JOBS TAG REWARDS: wrong results, right instruction counts for the rest).
#   usage: [LIBS="shipped lib4_skip_KILL ..."] tools/section_shares.sh "cfg3 tag5"      -> gpurun_out/sections.txt : VALU / SALU / LDS per wave-tick, cycles, launch us per variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
CFGS="${1:-cfg3 tag5}"
OUTF=gpurun_out/sections.txt
: > $OUTF
for CFG in $CFGS; do for LIB in ${LIBS:-shipped lib_skip_KILL lib_skip_JOBS lib_skip_TAG lib_skip_REWARDS}; do
  if [ "$LIB" = shipped ]; then unset SUSNET_LIB_PATH; else export SUSNET_LIB_PATH=$PWD/tools/_exp/$LIB.so; [ -f "$SUSNET_LIB_PATH" ] || continue; fi
  OUT=gpurun_out/sections/$CFG/$LIB
  rm -rf "$OUT" && mkdir -p "$OUT"
  BENCH="python3 bench.py --config $CFG --no-cpu-baseline --no-secondary --steps 10 --warmup 2 --repeats 0 --settle-ms 60"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- $BENCH > $OUT/pmc.log 2>&1 || { echo "$CFG $LIB pmc FAILED" >> $OUTF; continue; }
  $BENCH > $OUT/bench.json 2> $OUT/bench.err || { echo "$CFG $LIB bench FAILED" >> $OUTF; continue; }
  python3 - "$CFG" "$LIB" "$OUT" >> $OUTF <<'PY'
import csv, glob, json, sys
cfg, lib, out = sys.argv[1:4]
tot, n = {}, {}
for f in glob.glob(f"{out}/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_rollout" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
steps = n["SQ_WAVES"]
wt = tot["SQ_WAVES"] / steps * 512 * steps
b = json.loads(open(f"{out}/bench.json").read().strip().splitlines()[-1])
print(f"{cfg:5s} {lib:18s} VALU {tot['SQ_INSTS_VALU'] / wt:6.1f}  SALU {tot['SQ_INSTS_SALU'] / wt:5.1f}  LDS {tot['SQ_INSTS_LDS'] / wt:5.1f}  BR {tot['SQ_INSTS_BRANCH'] / wt:5.1f}  "
      f"cycles {4 * tot['SQ_WAVE_CYCLES'] / wt:6.0f}  launch_us {b['roofline']['avg_launch_us']:7.1f}  G {b['value'] / 1e9:6.1f}")
PY
done; done
cat $OUTF
