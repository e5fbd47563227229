#!/bin/bash
# Runs on the MI355X box from the repository root: the whole trainer step of the benchmark recipe with the decoder
# blocks' linear layers on the MFMA kernels where the measured table says so (the default) and all on the BLAS
# library, three runs each in turn on one box; prints ms per step of every run.
set -e -o pipefail
for round in 1 2 3; do
  for mode in auto lib; do
    PRL_LAYER_GEMMS=$mode timeout -k 10 300 python3 bench.py --workload trainer --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null \
      | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRL_LAYER_GEMMS=$mode', 'ms_per_step', round(d['ms_per_step'],2), 'loss', d['trainer_step'].get('loss'), 'grad_norm', d['trainer_step'].get('grad_norm'))"
  done
done
