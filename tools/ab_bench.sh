#!/bin/bash
# Same-box A/B of two builds of libe2etts_hip.so (boxes of the pool differ by several per cent, so numbers from two gpurun calls
# cannot rank builds).  Build both here (the second after `git stash`), copy them to tools/bin/lib_old.so and tools/bin/lib_new.so
# (git-ignored, shipped by gpurun), then on the GPU box:   bash tools/ab_bench.sh [kernel-class-regex]
# BENCH_ARGS (env): extra bench.py arguments, e.g. "--batch 1 --steps 20".
# Alternates old / new twice and prints ms/step of the matching classes and of the whole step.  Restores lib_new.so at the end.
set -e
pat=${1:-resblock_pair|conv_x3_128x128}
for v in old new old new; do
  cp tools/bin/lib_$v.so e2e_tts_amd/lib/libe2etts_hip.so
  python3 bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab.json 2> gpurun_out/ab.log
  echo "$v $(grep -E "\[bench\] ($pat)" gpurun_out/ab.log | awk '{printf "%s=%s ", $2, $6}') step=$(python3 -c 'import json;print(round(json.load(open("gpurun_out/ab.json"))["ms_per_step"],3))')"
done
cp tools/bin/lib_new.so e2e_tts_amd/lib/libe2etts_hip.so
