#!/bin/bash
# Kernel trace of the whole-trainer-step workload (MI355X box, from the repository root).
set -e -o pipefail
tag=${1:-trainer_prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trainer -- \
  python3 "$root/bench.py" --workload trainer --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_trainer.json" 2> "$out/trace.err"
cd "$root"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
find "$out/trace" -name "*kernel_trace.csv" -delete
echo done
