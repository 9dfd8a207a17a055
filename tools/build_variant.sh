#!/bin/bash
# Build a VARIANT of the library for same-box A/B runs (tools/ab_bench.sh) or diagnosis: tools/build_variant.sh NAME [extra hipcc flags]
#   -> tools/_exp/NAME.so (git-ignored; travels to the GPU box with gpurun).  Example: tools/build_variant.sh lib_stamps -DSUSNET_STAMPS
NAME="$1"; shift
SRC="${SUSNET_VARIANT_SRC:-$(dirname "$0")/../sus-net_amd/csrc}"
OBJ=$(mktemp -d)
mkdir -p "$(dirname "$0")/_exp"
for f in "$SRC"/*.hip; do
  case "$(basename "$f")" in inst_qnet*|inst_cfg2*|susnet_capi*) ILP="" ;; *) ILP="-mllvm -amdgpu-sched-strategy=max-ilp" ;; esac  # (as sus-net_amd/build_hip.py flags_for)
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-pass-failed $ILP "$@" -c -o "$OBJ/$(basename "$f" .hip).o" "$f" 2> "$OBJ/$(basename "$f").log" || { echo "FAILED $f"; tail -5 "$OBJ/$(basename "$f").log"; } ) &
  while [ "$(jobs -r | wc -l)" -ge 8 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$(dirname "$0")/_exp/$NAME.so" "$OBJ"/*.o && echo "built tools/_exp/$NAME.so"
rm -rf "$OBJ"
