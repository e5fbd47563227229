#!/bin/bash
# Kernel trace of the whole-trainer-step workload (MI355X box, from the repository root).
set -e -o pipefail
tag=${1:-trainer_prof}
impl=${2:-fused}  # hf | fused | replica | graphed
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trainer -- \
  python3 "$root/bench.py" --workload trainer --trainer-impl "$impl" --steps 2 --warmup 2 --no-cpu-baseline > "$out/bench_trainer.json" 2> "$out/trace.err"
cd "$root"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
find "$out/trace" -name "*kernel_trace.csv" -delete
echo done
