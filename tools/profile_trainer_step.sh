#!/bin/bash
# Runs on the MI355X box from the repository root: kernel trace of the whole trainer step of the benchmark recipe
# (8 timed steps); the per-kernel summary goes to gpurun_out/<tag>/kernel_stats.csv.
set -e -o pipefail
tag=${1:-trainer_prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trainer -- \
  python3 "$root/bench.py" --workload trainer --steps 8 --warmup 3 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
cd "$root"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
find "$out/trace" -name "*kernel_trace.csv" -delete
echo "profile done: $(ls "$out")"
