#!/bin/bash
# Kernel trace of the ragged-rollouts run (MI355X box, from the repository root).
set -e -o pipefail
root=$(pwd)
out=$root/gpurun_out/ragged_prof
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o ragged -- \
  python3 "$root/tools/ragged_rollouts_run.py" --steps 6 "$@" > "$out/run.log" 2> "$out/trace.err"
cd "$root"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
find "$out/trace" -name "*kernel_trace.csv" -delete
echo done
