#!/bin/bash
# Runs on the MI355X box from the repository root: a kernel trace of the benchmark step on the two-pass path
# (PRL_FUSED_ROWS=0: k_logprob_fwd + k_policy_small + k_logprob_bwd_once back to back, 20 steps), into gpurun_out/<tag>/.
set -e -o pipefail
tag=${1:-two_pass}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export PRL_FUSED_ROWS=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o tp -- \
  python3 "$root/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-trainer-extra --no-paths > "$out/bench.json" 2> "$out/trace.err"
cd "$root"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
rm -rf "$out/trace"
head -8 "$out/kernel_stats.csv" | cut -c1-150
