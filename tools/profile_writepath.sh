#!/bin/bash
# Runs on the GPU box (via gpurun): the write path of the headline command, one rocprofv3 --pmc pass per hardware block
# (never combined with other trace domains).  Output: gpurun_out/prof_$PROF_TAG/writepath/<group>/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
CFG=${1:-cfg2}
OUT=gpurun_out/prof_${PROF_TAG:-cur}/writepath_$CFG
[ -z "$ONLY_L2" ] && rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python3 bench.py --config $CFG --no-cpu-baseline --no-secondary --steps 6 --warmup 2"
# (every pass bounded and announced: a pass that stalls must not look like a hung box)
run() { name=$1; shift; echo "pass $name: $*"; timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1 || { echo "pass $name failed or timed out"; tail -3 $OUT/$name.log; }; }
if [ -z "$ONLY_L2" ]; then
run tcp TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCP_TA_ADDR_STALL_CYCLES
fi
# (the TCC / TA blocks take two or three counters per pass here: more "exceeds the capabilities of the hardware to collect")
run tcc1 TCC_EA0_WRREQ TCC_EA0_WRREQ_64B
run tcc2 TCC_WRITE TCC_REQ
run tcc3 TCC_EA0_WRREQ_STALL TCC_TAG_STALL
run ta1 TA_TA_BUSY
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
[ -z "$ONLY_L2" ] && run sq SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
echo writepath-ok
