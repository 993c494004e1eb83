#!/bin/bash
# Copy what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/<round>/ (tracked): bash tools/collect_profiles.sh r3p1 r3
# (the PMC summaries are made separately: tools/pmc_summary.py, see profiles/<round>/README.md)
set -e
src=gpurun_out/$1
dst=profiles/$2
mkdir -p "$dst"
cp "$src/bench.json" "$dst/bench_steps20_warmup5.json"
grep "\[bench\]" "$src/bench.err" > "$dst/bench_classes.txt"
for n in fp32 bf16x3 c3_mixed c3_mixed_bf16x3 b1_fp32 b1_bf16x3; do
  cp "$src/${n}_kernel_stats.csv" "$src/${n}_under_rocprof.json" "$src/${n}_classes.txt" "$dst/"
done
cp "$src/c5_kernel_stats.csv" "$src/c5_longform.txt" "$dst/"
ls -la "$dst"
