#!/bin/bash
# Ragged rollouts with half of the problems settled (advantage 0), 150 K tokens per step, 0.6B, 24 steps: the
# forward-only passes with the LM head at every position (PRL_GATHER_SCORED=0, which also switches the gathering off
# in the training passes) against the default; per arm the steps' times (the first two left out), their sum and median.
#   bash tools/settled_ab.sh > gpurun_out/settled_ab.txt
set -e
ARGS="--seq-len 512 --micro-batch 32 --batch 512 --steps 24 --settled 0.5"
summarise() {
  python3 -c "
import ast, statistics, sys
t = [ast.literal_eval(l)['time/step'] for l in sys.stdin if l.startswith('{')][2:]
print(' '.join(f'{x:.3f}' for x in t)); print(f'sum {sum(t):.3f} s  median {statistics.median(t):.3f} s over {len(t)} steps')"
}
for arm in default PRL_GATHER_SCORED=0 default PRL_GATHER_SCORED=0; do
  echo "== $arm"
  if [ "$arm" = default ]; then
    timeout -k 10 240 python3 -u tools/ragged_rollouts_run.py $ARGS 2>&1 | summarise
  else
    env "$arm" timeout -k 10 240 python3 -u tools/ragged_rollouts_run.py $ARGS 2>&1 | summarise
  fi
done
