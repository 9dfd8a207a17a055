#!/bin/bash
# A/B on the GPU box: tools/ab_bench.sh "<lib1.so|shipped> <lib2.so> ..." "<cfg ...>" [extra bench.py flags]
# prints one line per (library, config, repetition): env-steps/s (G), launch us, episodes, episode steps
LIBS="$1"; CFGS="$2"; shift 2
mkdir -p gpurun_out/ab
for c in $CFGS; do for lib in $LIBS; do for rep in 1 2; do
  if [ "$lib" = shipped ]; then unset SUSNET_LIB_PATH; else export SUSNET_LIB_PATH=$PWD/tools/_exp/$lib; fi
  timeout -k 10 150 python bench.py --config $c --steps 20 --warmup 5 --no-secondary --no-cpu-baseline "$@" > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err || { echo "$lib $c FAILED"; tail -3 gpurun_out/ab/b.err; continue; }
  python - "$lib" "$c" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab/b.json"))
m = d["episode_metrics"]
print(sys.argv[1], sys.argv[2], round(d["value"] / 1e9, 2), "G", round(d["roofline"]["avg_launch_us"], 1), "us", "episodes", m.get("episodes"), "steps", m.get("episode_steps"))
PY
done; done; done
