#!/bin/bash
# Runs on the GPU box (via gpurun): the policy-in-the-loop configuration (BASELINE config 5) under rocprofv3 --kernel-trace --stats,
# (the policy loop in blocks of 5 ticks per launch -- bench.py's default --policy-block: the reference trainer's train_step_interval --, with one launch per tick and in blocks of 64), unprofiled beside it, and the network kernel's own timing (tools/qnet_bench.py).  Output under gpurun_out/prof_$PROF_TAG/cfg5/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_${PROF_TAG:-cur}/cfg5
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="bench.py --config cfg5 --steps 256 --warmup 32 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $CMD > $OUT/bench_under_rocprofv3.json 2> $OUT/stats.err || exit 1
echo "profiled cfg5"
python3 $CMD > $OUT/bench_unprofiled.json 2> $OUT/unprofiled.err || exit 1
# the same loop with ONE launch per tick (susnet_qnet_policy_step): what a caller that cannot batch ticks gets
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_per_tick -- python3 $CMD --policy-block 0 > $OUT/bench_per_tick_under_rocprofv3.json 2> $OUT/stats_per_tick.err || exit 1
find $OUT/stats_per_tick -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_per_tick.csv
# ... and with 64 ticks per launch (run_game's loop with fixed networks: round 4's headline form)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_block64 -- python3 $CMD --policy-block 64 > $OUT/bench_block64_under_rocprofv3.json 2> $OUT/stats_block64.err || exit 1
find $OUT/stats_block64 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_block64.csv
python3 tools/policy_block_bench.py > $OUT/policy_block_bench.json 2> $OUT/policy_block.err || exit 1
python3 tools/qnet_bench.py > $OUT/qnet_bench.json 2> $OUT/qnet.err || exit 1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
echo cfg5-profile-ok
# the network kernel's own counters: MFMA instructions, matrix-busy cycles, VALU / LDS / VMEM instruction counts, memory traffic
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_mfma -- python3 tools/qnet_bench.py > $OUT/pmc_mfma.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_inst -- python3 tools/qnet_bench.py > $OUT/pmc_inst.log 2>&1 || exit 1
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 tools/qnet_bench.py > $OUT/pmc_$C.log 2>&1 || exit 1
done
echo cfg5-pmc-ok
