#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: bench line, rocprofv3 kernel trace and the three PMC passes that
# tools/pmc_summary.py reads.  Everything lands under gpurun_out/$1/ (scratch); copy the summaries into profiles/ afterwards:
#   python tools/pmc_summary.py gpurun_out/$1 profiles/<round>
set -e -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
echo "[profile] bench done"
rocprofv3 --kernel-trace --stats -d "$out/trace" --output-format csv -- python3 bench.py --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
echo "[profile] kernel trace done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=${pass%% *}
  rocprofv3 --pmc $pass -d "$out/$name" --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/$name.json" 2> "$out/$name.err"
  echo "[profile] pmc pass $name done"
done
