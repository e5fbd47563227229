#!/bin/bash
# Runs on the MI355X box from the repository root: kernel trace of a short bench run, then the timeline of its end.
set -e -o pipefail
root=$(pwd)
out=$root/gpurun_out/timeline
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o bench -- \
  python3 "$root/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-trainer-extra "$@" > "$out/bench.json" 2> "$out/trace.err"
cd "$root"
python3 tools/step_timeline.py "$out/trace" ${TIMELINE_KERNELS:-400} > "$out/timeline.txt"
find "$out/trace" -name "*kernel_trace.csv" -delete
tail -30 "$out/timeline.txt"
