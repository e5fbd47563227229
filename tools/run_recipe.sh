#!/bin/bash
# Runs on the MI355X box: one recipe through the trainer entry point on one rank; the log goes to gpurun_out/.
# usage: tools/run_recipe.sh <recipe.toml> [extra --flags]
set -o pipefail
recipe=$1; shift
name=$(basename "$recipe" .toml)
mkdir -p gpurun_out
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 LOCAL_WORLD_SIZE=1 \
  timeout -k 10 500 python -u -m prime_rl_cse_amd.train @ "$recipe" --output-dir /tmp/prl_out_$name "$@" \
  > gpurun_out/recipe_$name.log 2>&1
rc=$?
grep -v "Warning\|warn" gpurun_out/recipe_$name.log | tail -12 | cut -c1-300
exit $rc
