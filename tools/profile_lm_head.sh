#!/bin/bash
# Runs on the MI355X box from the repository root: a kernel trace and two counter passes of a program that
# launches only the LM head's three products (this package's kernels and the library's), into gpurun_out/<tag>/.
set -e -o pipefail
tag=${1:-lm_head}
shape=${2:-"16384 1024 151936"}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o lm -- \
  python3 "$root/tools/lm_head_only.py" 6 $shape both > "$out/under_trace.txt" 2> "$out/trace.err"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/pmc_a" -o pmc -- python3 "$root/tools/lm_head_only.py" 3 $shape both > "$out/pmc_a.txt" 2> "$out/pmc_a.err"
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS FETCH_SIZE GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/pmc_b" -o pmc -- python3 "$root/tools/lm_head_only.py" 3 $shape both > "$out/pmc_b.txt" 2> "$out/pmc_b.err"
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum \
  --output-format csv -d "$out/pmc_c" -o pmc -- python3 "$root/tools/lm_head_only.py" 3 $shape both > "$out/pmc_c.txt" 2> "$out/pmc_c.err"
cd "$root"
for k in "k_lm_gemm<true, 0>" "k_lm_gemm<false, 0>" "k_lm_gemm<false, 1>" "k_lm_gemm<false, 2>" "Cijk"; do
  name=$(echo "$k" | tr -c 'A-Za-z0-9' '_')
  python3 tools/pmc_rows_summary.py "$k" "$out/pmc_$name.json" "$out/pmc_a" "$out/pmc_b" "$out/pmc_c" > /dev/null || true
done
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
find "$out/pmc_a" -name "*counter_collection.csv" -exec cp {} "$out/pmc_a_counters.csv" \;
rm -rf "$out/pmc_b" "$out/pmc_c" "$out/trace" "$out/pmc_a"
echo "lm head profile done: $(ls "$out")"
