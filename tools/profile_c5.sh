#!/bin/bash
# BASELINE config 5 (48 kHz long-form streaming vocoder) under the profiler; run on the GPU box from the repo root:
#   bash tools/profile_c5.sh <tag> [modes, default bf16] [chunk sizes, default "512 5632"]
# Per chunk size: plain timing, the per-layer event profile (one kernel at a time), a rocprofv3 kernel trace (stats + the timeline of one
# chunk) and four counter passes (--pmc alone, no trace domains; the program itself follows `--`).  Everything lands under
# gpurun_out/<tag>/; summarise with  python3 tools/pmc_summary.py gpurun_out/<tag>/pmc_<chunk> profiles/<round> --name pmc_c5_<chunk>.md
set -e -o pipefail
tag=${1:-c5}
modes=${2:-bf16}
chunks=${3:-512 5632}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for ch in $chunks; do
  python3 tools/longform_bench.py $ch $modes 5 > "$out/time_$ch.txt" 2> "$out/time_$ch.err"
  E2ETTS_PROFILE_FINE=1 python3 tools/longform_bench.py $ch $modes 1 > "$out/fine_$ch.txt" 2> "$out/fine_$ch.err"
  echo "[profile_c5] chunk $ch timing done: $(head -1 "$out/time_$ch.txt")"
  rocprofv3 --kernel-trace --stats -d "$out/trace_$ch" --output-format csv -- python3 tools/longform_bench.py $ch $modes 2 > "$out/under_rocprof_$ch.txt" 2> "$out/trace_$ch.err"
  cp "$(find "$out/trace_$ch" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_$ch.csv"
  python3 tools/b1_timeline.py "$(find "$out/trace_$ch" -name '*kernel_trace.csv' | head -1)" "conv_post_kernel<256>" $([ "$ch" -le 1024 ] && echo "4 2" || echo "1 1") > "$out/timeline_$ch.txt"
  rm -rf "$out/trace_$ch"
  echo "[profile_c5] chunk $ch kernel trace done"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
    name=${pass%% *}
    rocprofv3 --pmc $pass -d "$out/pmc_$ch/$name" --output-format csv -- python3 tools/longform_bench.py $ch $modes 1 > "$out/pmc_${ch}_$name.txt" 2> "$out/pmc_${ch}_$name.err" \
      || echo "[profile_c5] pmc pass $name FAILED (see $out/pmc_${ch}_$name.err)"
    echo "[profile_c5] chunk $ch pmc pass $name done"
  done
done
