#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 passes for the driver-shaped bench command of each config.
#   usage: [PROF_TAG=r02x] tools/profile_gpu.sh [cfg2 cfg3 cfg4 tag5 ...]      (default: all four)
# Output under gpurun_out/prof_$PROF_TAG/<config>/ (a fresh directory per tag: gpurun MERGES into the local gpurun_out/).  --kernel-trace --stats in one run; each PMC group in its own run (never
# combined with other trace domains).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
CONFIGS="${*:-cfg2 cfg2w cfg3 cfg4 tag5}"
for CFG in $CONFIGS; do
  OUT=gpurun_out/prof_${PROF_TAG:-cur}/$CFG
  rm -rf "$OUT" && mkdir -p "$OUT"
  BENCH="python3 bench.py --config $CFG --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --repeats 0 --settle-ms 150"
  if [ "$CFG" = cfg2 ]; then FULL="python3 bench.py --config cfg2 --no-cpu-baseline --steps 20 --warmup 5 --repeats 0 --settle-ms 150"; else FULL="$BENCH"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $FULL > $OUT/stats.log 2>&1 || exit 1
  for C in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- $BENCH > $OUT/pmc_$C.log 2>&1 || exit 1
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 1
  if [ "$CFG" = cfg2 ]; then
    for O in planes flat; do
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_W_$O -- python3 bench.py --obs $O --ticks 128 --no-cpu-baseline --no-secondary --steps 4 --warmup 1 > $OUT/pmc_W_$O.log 2>&1 || exit 1
    done
  fi
  echo "profiled $CFG"
done
echo profiles-ok
