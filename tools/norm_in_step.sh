#!/bin/bash
# Runs on the MI355X box from the repository root: the whole trainer step of the benchmark recipe with the gradient
# norm taken from the accumulation kernel's per-chunk sums of squares (the default) and from a pass over the finished
# sums, three runs each in turn on one box; prints ms per step of every run.
set -e -o pipefail
for round in 1 2 3; do
  for mode in 1 0; do
    PRL_NORM_FROM_CHUNKS=$mode timeout -k 10 300 python3 bench.py --workload trainer --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null \
      | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRL_NORM_FROM_CHUNKS=$mode', 'ms_per_step', round(d['ms_per_step'],2))"
  done
done
