#!/bin/bash
# Runs on the MI355X box from the repository root: kernel trace of the default bench workload, then the two
# HBM counter passes (each in a run of its own), everything under gpurun_out/<tag>/.
set -e -o pipefail
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- \
  python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-trainer-extra --no-paths > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o pmc -- \
  python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-trainer-extra --no-paths --no-stats > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o pmc -- \
  python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-trainer-extra --no-paths --no-stats > "$out/pmc_write.json" 2> "$out/pmc_write.err"
cd "$root"
python3 tools/pmc_summary.py "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_traffic.json" > /dev/null
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
rm -rf "$out/pmc_fetch" "$out/pmc_write"
find "$out/trace" -name "*kernel_trace.csv" -delete
echo "profile done: $(ls "$out")"
