#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: the bench line with the driver's flags, rocprofv3 kernel traces of the same
# command per workload -- the headline (exact fp32), the split-precision fast mode, BASELINE config 3 (mixed lengths), config 5 (48 kHz
# long-form streaming) and config 2 (B = 1) -- and the three PMC passes that tools/pmc_summary.py reads, for the headline, the fast mode
# and config 3.  Everything lands under gpurun_out/$1/ (scratch); copy the summaries into profiles/ afterwards:
#   python tools/pmc_summary.py gpurun_out/$1 profiles/<round>
# (counter passes use --pmc alone, no trace domains; the program itself follows `--`, never a shell or env wrapper)
set -e -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
echo "[profile] bench done"
trace() {  # name, then the bench.py arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d "$out/trace_$name" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras "$@" > "$out/${name}_under_rocprof.json" 2> "$out/trace_$name.err"
  cp "$(find "$out/trace_$name" -name '*kernel_stats.csv' | head -1)" "$out/${name}_kernel_stats.csv"
  grep "\[bench\]" "$out/trace_$name.err" > "$out/${name}_classes.txt" || true
  rm -rf "$out/trace_$name"
  echo "[profile] kernel trace $name done"
}
trace fp32
trace bf16x3 --precision bf16x3
trace c3_mixed --workload mixed
trace c3_mixed_bf16x3 --workload mixed --precision bf16x3
trace b1_fp32 --batch 1 --steps 50 --warmup 5
trace b1_bf16x3 --batch 1 --steps 50 --warmup 5 --precision bf16x3
rocprofv3 --kernel-trace --stats -d "$out/trace_c5" --output-format csv -- python3 tools/longform_bench.py > "$out/c5_longform.txt" 2> "$out/trace_c5.err"
cp "$(find "$out/trace_c5" -name '*kernel_stats.csv' | head -1)" "$out/c5_kernel_stats.csv"
rm -rf "$out/trace_c5"
echo "[profile] kernel trace c5 done"
for mode in fp32 bf16x3 c3; do
  args="--precision $mode"
  [ $mode = c3 ] && args="--workload mixed"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    name=${pass%% *}_$mode
    [ $mode = c3 ] && name=c3/$name && mkdir -p "$out/c3"   # config 3's counters in a directory of their own: tools/pmc_summary.py sums per class over a tree
    rocprofv3 --pmc $pass -d "$out/$name" --output-format csv -- python3 bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$out/$name.json" 2> "$out/$name.err"
    echo "[profile] pmc pass $name done"
  done
done
