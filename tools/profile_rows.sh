#!/bin/bash
# Runs on the MI355X box from the repository root: everything that is known about where k_policy_rows' time goes,
# into gpurun_out/<tag>/ : variants and phase trace (tune_rows.py), the load/store skeletons (rowcopy_bench), a
# kernel trace and two counter passes of a program that launches only this kernel.
# Build first, in the build container: python tools/tune_rows.py build; hipcc ... tools/rowcopy_bench.hip
set -e -o pipefail
tag=${1:-rows}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 200 python3 tools/tune_rows.py run > "$out/tune_rows.txt" 2> "$out/tune_rows.err"
if [ -x tools/bin/rowcopy_bench ]; then timeout -k 10 100 tools/bin/rowcopy_bench > "$out/rowcopy_bench.txt" 2>&1; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o rows -- \
  python3 "$root/tools/rows_only.py" 48 > "$out/rows_only_under_trace.txt" 2> "$out/trace.err"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM \
  --output-format csv -d "$out/pmc_a" -o pmc -- python3 "$root/tools/rows_only.py" 12 > "$out/pmc_a.txt" 2> "$out/pmc_a.err"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CU_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/pmc_b" -o pmc -- python3 "$root/tools/rows_only.py" 12 > "$out/pmc_b.txt" 2> "$out/pmc_b.err"
cd "$root"
python3 tools/pmc_rows_summary.py k_policy_rows "$out/pmc_rows.json" "$out/pmc_a" "$out/pmc_b" > /dev/null
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
rm -rf "$out/pmc_a" "$out/pmc_b" "$out/trace"
echo "rows profile done: $(ls "$out")"
