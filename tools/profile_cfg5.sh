#!/bin/bash
# Runs on the GPU box (via gpurun): the policy-in-the-loop configuration (BASELINE config 5) under rocprofv3 --kernel-trace --stats,
# unprofiled beside it, and the network kernel's own timing (tools/qnet_bench.py).  Output under gpurun_out/prof_$PROF_TAG/cfg5/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_${PROF_TAG:-cur}/cfg5
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="bench.py --config cfg5 --steps 256 --warmup 32 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $CMD > $OUT/bench_under_rocprofv3.json 2> $OUT/stats.err || exit 1
echo "profiled cfg5"
python3 $CMD > $OUT/bench_unprofiled.json 2> $OUT/unprofiled.err || exit 1
python3 tools/qnet_bench.py > $OUT/qnet_bench.json 2> $OUT/qnet.err || exit 1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
echo cfg5-profile-ok
