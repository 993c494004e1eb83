#!/bin/bash
# Copy what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/<round>/ (tracked) and summarise its counter passes:
#   bash tools/collect_profiles.sh r4p1 r4
# Only what exists is taken, so it also serves a --only run.
set -e
src=gpurun_out/$1
dst=profiles/$2
mkdir -p "$dst"
if [ -f "$src/bench.json" ]; then
  cp "$src/bench.json" "$dst/bench_steps20_warmup5.json"
  grep "\[bench\]" "$src/bench.err" > "$dst/bench_classes.txt" || true
fi
for n in fp32 bf16x3 c3_mixed c3_mixed_bf16x3 b1_fp32 b1_bf16x3; do
  for f in "$src/${n}_kernel_stats.csv" "$src/${n}_under_rocprof.json" "$src/${n}_classes.txt" "$src/${n}_timeline.txt"; do
    [ -f "$f" ] && cp "$f" "$dst/" || true
  done
done
[ -d "$src/pmc_headline" ] && python3 tools/pmc_summary.py "$src/pmc_headline" "$dst" --title "Round ${2#r}, headline workload (B = 32 x L = 128): exact fp32 and the bf16x3 fast mode"
[ -d "$src/c3" ] && python3 tools/pmc_summary.py "$src/c3" "$dst" --name pmc_c3_mixed.md --args "--workload mixed" --title "Round ${2#r}, BASELINE config 3 (B = 32, lengths 40-200, ragged compute), exact fp32"
if [ -d "$src/c5" ]; then
  for ch in 512 5632; do
    [ -f "$src/c5/time_$ch.txt" ] || continue
    cp "$src/c5/time_$ch.txt" "$dst/c5_time_$ch.txt"
    cp "$src/c5/fine_$ch.txt" "$dst/c5_layers_$ch.txt"
    cp "$src/c5/kernel_stats_$ch.csv" "$dst/c5_kernel_stats_$ch.csv"
    cp "$src/c5/timeline_$ch.txt" "$dst/c5_timeline_$ch.txt"
    extra=""
    [ $ch = 512 ] && extra="--traffic pmc_traffic_c5.json"
    python3 tools/pmc_summary.py "$src/c5/pmc_$ch" "$dst" --name pmc_c5_$ch.md $extra --cmd "python3 tools/longform_bench.py $ch bf16 1" \
      --title "Round ${2#r}, BASELINE config 5 (48 kHz HiFi-GAN, 60.07 s, plain bf16), mel pushed in chunks of $ch frames"
  done
  # the name VERDICT r3 asked for: the chunk-512 table (the configuration the bench line quotes), with a pointer to the one-call table
  if [ -f "$dst/pmc_c5_512.md" ]; then
    { cat "$dst/pmc_c5_512.md"; echo; echo "(One call on the whole utterance: \`pmc_c5_5632.md\`.  Round 3's kernels under the same passes: \`pmc_c5_512_round3_build.md\`.)"; } > "$dst/pmc_c5.md"
  fi
fi
ls -la "$dst"
