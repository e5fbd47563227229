#!/bin/bash
# Runs on the MI355X box from the repository root: the other recipes' trainer steps (through the entry point) with one
# environment switch at two settings, in turn, twice each; prints ms per step of every run.
#   tools/recipes_switch_ab.sh PRL_LAYER_GEMMS auto lib [recipe ...]
set -e -o pipefail
var=$1; a=$2; b=$3; shift 3
recipes=${@:-wordle hendrycks_7b}
for recipe in $recipes; do
  for round in 1 2; do
    for mode in $a $b; do
      env $var=$mode timeout -k 10 400 python3 bench.py --recipe $recipe --steps 3 --warmup 2 2>/dev/null \
        | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$recipe $var=$mode', round(d['ms_per_step'],1))"
    done
  done
done
