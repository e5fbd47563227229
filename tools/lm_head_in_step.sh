#!/bin/bash
# Runs on the MI355X box from the repository root: the whole trainer step of the benchmark recipe with the LM head's
# backward products on this package's kernels (the default) and with all three on the BLAS library, two runs each in
# turn on one box; prints ms per step of every run.
set -e -o pipefail
for round in 1 2; do
  for mode in auto lib; do
    PRL_LM_HEAD=$mode timeout -k 10 300 python3 bench.py --workload trainer --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null \
      | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRL_LM_HEAD=$mode', 'ms_per_step', round(d['ms_per_step'],2))"
  done
done
