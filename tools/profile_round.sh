#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: the bench line, a rocprofv3 kernel trace of the same command, and the three PMC
# passes that tools/pmc_summary.py reads -- for the headline arithmetic (exact fp32) and for the split-precision fast mode (bf16x3).
# Everything lands under gpurun_out/$1/ (scratch); copy the summaries into profiles/ afterwards:
#   python tools/pmc_summary.py gpurun_out/$1 profiles/<round>
set -e -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
echo "[profile] bench done"
rocprofv3 --kernel-trace --stats -d "$out/trace" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
echo "[profile] kernel trace (fp32) done"
rocprofv3 --kernel-trace --stats -d "$out/trace_x3" --output-format csv -- python3 bench.py --precision bf16x3 --no-cpu-baseline --no-extras > "$out/bench_x3_under_rocprof.json" 2> "$out/trace_x3.err"
echo "[profile] kernel trace (bf16x3) done"
for mode in fp32 bf16x3; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    name=${pass%% *}_$mode
    rocprofv3 --pmc $pass -d "$out/$name" --output-format csv -- python3 bench.py --precision $mode --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$out/$name.json" 2> "$out/$name.err"
    echo "[profile] pmc pass $name done"
  done
done
