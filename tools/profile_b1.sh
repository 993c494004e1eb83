#!/bin/bash
# The B = 1 latency path under rocprofv3 (kernel trace + stats), both arithmetic modes; run on the GPU box from the repo root:
#   bash tools/profile_b1.sh <tag>     -> gpurun_out/<tag>/b1_{fp32,bf16x3}.json, b1_*_kernel_stats.csv, b1_*_classes.txt
set -e -o pipefail
tag=${1:-b1}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for pr in fp32 bf16x3; do
  python3 bench.py --precision $pr --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$out/b1_$pr.json" 2> "$out/b1_$pr.err"
  grep "\[bench\]" "$out/b1_$pr.err" > "$out/b1_${pr}_classes.txt"
  rocprofv3 --kernel-trace --stats -d "$out/trace_$pr" --output-format csv -- python3 bench.py --precision $pr --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$out/b1_${pr}_under_rocprof.json" 2> "$out/trace_$pr.err"
  cp "$(find "$out/trace_$pr" -name '*kernel_stats.csv' | head -1)" "$out/b1_${pr}_kernel_stats.csv"
  echo "[profile_b1] $pr done: $(python3 -c "import json;d=json.load(open('$out/b1_$pr.json'));print(d['ms_per_step_median'])") ms/step"
done
