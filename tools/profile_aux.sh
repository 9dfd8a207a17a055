#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + WRITE_SIZE for the feature / replay-ring kernels
# (susnet_featurize, susnet_ring_append).  Output under gpurun_out/prof_$PROF_TAG/{featurize,ring}/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for W in featurize ring; do
  OUT=gpurun_out/prof_${PROF_TAG:-cur}/$W
  rm -rf "$OUT" && mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/${W}_bench.py > $OUT/stats.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_WRITE_SIZE -- python3 tools/${W}_bench.py > $OUT/pmc_WRITE_SIZE.log 2>&1 || exit 1
  python3 tools/${W}_bench.py > $OUT/unprofiled.log 2>&1 || exit 1
  echo "profiled $W"
done
echo aux-profiles-ok
