#!/bin/bash
# Runs on the MI355X box from the repository root: the single-pass row kernel's time INSIDE the whole trainer step
# (right behind the LM head's product, which has just written the 5 GB of logits), with non-temporal gradient
# stores (the default) and with plain ones (a diagnostic build), from two kernel traces; into gpurun_out/<tag>/.
set -e -o pipefail
tag=${1:-rows_in_step}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in default plain_store; do
  if [ $v = plain_store ]; then export PRL_HIP_LIB=$root/prime-rl-cse_amd/lib/variants/libprl_rows_plain_store.so; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$v" -o trainer -- \
    python3 "$root/bench.py" --workload trainer --steps 4 --warmup 2 --no-cpu-baseline > "$out/bench_$v.json" 2> "$out/trace_$v.err"
  find "$out/trace_$v" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_$v.csv" \;
  rm -rf "$out/trace_$v"
  echo "== $v"; grep -E "k_policy_rows|k_lm_gemm|Cijk_Alik_Bljk_BBS_BH_Bias_HA_S_SAV_NTD|ms_per_step" "$out/kernel_stats_$v.csv" | cut -c1-60,200-400 | head -5
  python3 -c "
import json,sys
d=json.loads(open('$out/bench_$v.json').read().strip().splitlines()[-1]); print('ms_per_step', d.get('ms_per_step'))"
done
