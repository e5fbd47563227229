# instruction-cache and issue counters of the LM-head kernels for a given library build (PRL_HIP_LIB), into gpurun_out/<tag>/
set -e -o pipefail
tag=$1
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
if [ -n "$PRL_HIP_LIB" ]; then export PRL_HIP_LIB="$root/$PRL_HIP_LIB"; fi
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/pmc" -o pmc -- python3 "$root/tools/lm_head_only.py" 3 16384 1024 151936 kernel > "$out/pmc.txt" 2> "$out/pmc.err"
cd "$root"
python3 tools/pmc_kernel_means.py "$(find "$out/pmc" -name "*counter_collection.csv" | head -1)" lm_gemm | tee "$out/means.txt"
rm -rf "$out/pmc"
