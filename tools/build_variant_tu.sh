#!/bin/bash
# Build a VARIANT of the library that differs from the shipped one in ONE translation unit (seconds instead of minutes):
#   tools/build_variant_tu.sh NAME inst_cfg4 [extra hipcc flags]  -> tools/_exp/NAME.so = the shipped objects (sus-net_amd/_obj, must be
#   current: build the shipped library first) with that unit recompiled under the extra flags.  For diagnostic builds of one kernel family
#   (SUSNET_EXP_* macros) and same-box A/B runs (tools/ab_bench.sh).
NAME="$1"; TU="$2"; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OBJ=$(mktemp -d)
mkdir -p "$ROOT/tools/_exp"
case "$TU" in inst_qnet*|inst_cfg2*|susnet_capi*) ILP="" ;; *) ILP="-mllvm -amdgpu-sched-strategy=max-ilp" ;; esac  # (as sus-net_amd/build_hip.py flags_for)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-pass-failed $ILP "$@" -c -o "$OBJ/$TU.o" "$ROOT/sus-net_amd/csrc/$TU.hip" || { echo "FAILED $TU"; exit 1; }
OTHERS=$(ls "$ROOT"/sus-net_amd/_obj/*.o | grep -v "/$TU.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/_exp/$NAME.so" "$OBJ/$TU.o" $OTHERS && echo "built tools/_exp/$NAME.so"
rm -rf "$OBJ"
