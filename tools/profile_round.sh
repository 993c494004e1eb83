#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: the bench line with the driver's flags, rocprofv3 kernel traces of the same
# command per workload, and the PMC passes that tools/pmc_summary.py reads.  Everything lands under gpurun_out/<tag>/ (scratch); copy the
# summaries into profiles/ afterwards: bash tools/collect_profiles.sh <tag> <round> (copies the traces and runs tools/pmc_summary.py on
# pmc_headline/, c3/ and c5/pmc_512/).
#
#   bash tools/profile_round.sh <tag>                      the whole set -- ONCE per round, after the last commit that touches csrc/
#   bash tools/profile_round.sh <tag> --only headline      bench line + fp32 trace + fp32 PMC passes           (the headline workload)
#   bash tools/profile_round.sh <tag> --only bf16x3        split-precision trace + PMC passes
#   bash tools/profile_round.sh <tag> --only c3            config 3 (mixed lengths): traces (fp32, bf16x3) + PMC passes
#   bash tools/profile_round.sh <tag> --only b1            config 2 (B = 1): traces in both arithmetic modes + the timeline of one step
#   bash tools/profile_round.sh <tag> --only c5            config 5 (48 kHz stream): tools/profile_c5.sh (timing, per-layer table, trace, PMC)
# An experiment re-measures ONE workload with --only; commits that only touch code the benchmark does not run do not re-take profiles/.
# (counter passes use --pmc alone, no trace domains; the program itself follows `--`, never a shell or env wrapper)
set -e -o pipefail
tag=${1:-prof}
only=all
[ "$2" = "--only" ] && only=${3:?--only needs a workload: headline | bf16x3 | c3 | b1 | c5}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
want() { [ "$only" = all ] || [ "$only" = "$1" ]; }
trace() {  # name, then the bench.py arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d "$out/trace_$name" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras "$@" > "$out/${name}_under_rocprof.json" 2> "$out/trace_$name.err"
  cp "$(find "$out/trace_$name" -name '*kernel_stats.csv' | head -1)" "$out/${name}_kernel_stats.csv"
  grep "\[bench\]" "$out/trace_$name.err" > "$out/${name}_classes.txt" || true
  if [ "${name#b1_}" != "$name" ]; then   # the B = 1 path: order and overlap of ~190 short launches
    python3 tools/b1_timeline.py "$(find "$out/trace_$name" -name '*kernel_trace.csv' | head -1)" > "$out/${name}_timeline.txt" || true
  fi
  rm -rf "$out/trace_$name"
  echo "[profile] kernel trace $name done"
}
pmc() {  # directory name under $out, then the bench.py arguments
  local dir=$1; shift
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    local name=${pass%% *}
    mkdir -p "$out/$dir"
    rocprofv3 --pmc $pass -d "$out/$dir/$name" --output-format csv -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$out/$dir/$name.json" 2> "$out/$dir/$name.err"
    echo "[profile] pmc pass $dir/$name done"
  done
}
if want headline; then
  python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
  echo "[profile] bench done"
  trace fp32
  pmc pmc_headline/fp32 --precision fp32
fi
if want bf16x3; then
  trace bf16x3 --precision bf16x3
  pmc pmc_headline/bf16x3 --precision bf16x3
fi
if want c3; then
  trace c3_mixed --workload mixed
  trace c3_mixed_bf16x3 --workload mixed --precision bf16x3
  pmc c3 --workload mixed
fi
if want b1; then
  trace b1_fp32 --batch 1 --steps 50 --warmup 5
  trace b1_bf16x3 --batch 1 --steps 50 --warmup 5 --precision bf16x3
fi
if want c5; then
  bash tools/profile_c5.sh "$tag/c5" bf16 "512 5632"
fi
