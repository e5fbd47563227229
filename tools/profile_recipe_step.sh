#!/bin/bash
# Runs on the MI355X box from the repository root: kernel trace of another recipe's trainer step through the entry
# point (bench.py --recipe NAME, 2 warm-up + 3 timed steps); the per-kernel summary goes to
# gpurun_out/<tag>/kernel_stats.csv.      tools/profile_recipe_step.sh wordle [tag]
set -e -o pipefail
recipe=${1:-wordle}
tag=${2:-recipe_prof_$recipe}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o recipe -- \
  python3 "$root/bench.py" --recipe "$recipe" --steps 3 --warmup 2 > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
cd "$root"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
find "$out/trace" -name "*kernel_trace.csv" -delete
echo "profile done: $(ls "$out")"
