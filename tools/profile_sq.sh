#!/bin/bash
# Runs on the GPU box (via gpurun): only the two SQ counter passes of tools/profile_gpu.sh (instruction mix and issue / wait cycles
# per wave), for quick looks between kernel changes.   usage: PROF_TAG=r04a tools/profile_sq.sh cfg3 cfg4 tag5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for CFG in "$@"; do
  OUT=gpurun_out/prof_${PROF_TAG:-cur}/$CFG
  rm -rf "$OUT" && mkdir -p "$OUT"
  BENCH="python3 bench.py --config $CFG --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --repeats 0 --settle-ms 150"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 1
  echo "profiled $CFG"
done
python3 - "$PROF_TAG" "$@" <<'PY'
import collections, csv, glob, sys
tag = sys.argv[1]
for cfg in sys.argv[2:]:
    tot = collections.defaultdict(list)
    for grp in ("pmc_sq1", "pmc_sq2"):
        for f in glob.glob(f"gpurun_out/prof_{tag}/{cfg}/{grp}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "k_rollout" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in tot.items()}
    if not m: continue
    n = len(tot["SQ_WAVES"])  # 33 bench steps of 512 ticks (5 warm-up + 20 timed + 8 event pairs); a step split into consecutive launches holds fewer ticks each
    wt = m["SQ_WAVES"] * (512 * 33 / n if n % 33 == 0 else 512)  # wave-ticks per launch
    cyc = m["SQ_WAVE_CYCLES"] * 4 / wt
    print(cfg, "per wave-tick:", " ".join(f"{k[3:]}={m[k] / wt:.1f}" for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")),
          f"cycles={cyc:.0f}", " ".join(f"{k[3:]}={m[k] * 4 / wt:.0f}" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC") if k in m))
PY
