#!/bin/bash
# Same-box A/B of builds of libe2etts_hip.so (boxes of the pool differ by several per cent, so numbers from two gpurun calls cannot rank
# builds).  Build the variants here, copy them to tools/bin/lib_<name>.so (git-ignored, shipped by gpurun), then on the GPU box:
#   VARIANTS="old new" bash tools/ab_bench.sh [kernel-class-regex]
# BENCH_ARGS (env): extra bench.py arguments, e.g. "--precision bf16x3 --batch 1 --steps 20".
# Runs the variant list twice and prints ms/step of the matching classes and of the whole step.  Restores the LAST variant at the end.
set -e
pat=${1:-resblock_pair|conv_x3_128x128}
vars=${VARIANTS:-old new}
for v in $vars $vars; do
  cp tools/bin/lib_$v.so e2e_tts_amd/lib/libe2etts_hip.so
  python3 bench.py --no-cpu-baseline --no-extras $BENCH_ARGS > gpurun_out/ab.json 2> gpurun_out/ab.log
  echo "$v $(grep -E "\[bench\] ($pat)" gpurun_out/ab.log | awk '{printf "%s=%s ", $2, $6}') step=$(python3 -c 'import json;print(round(json.load(open("gpurun_out/ab.json"))["ms_per_step"],3))')"
done
