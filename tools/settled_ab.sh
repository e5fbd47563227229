#!/bin/bash
# Ragged rollouts with half of the problems settled (advantage 0), 150 K tokens per step, 0.6B, 24 steps, through the
# trainer entry point: the default against the same run with one environment variable set (default: PRL_GATHER_SCORED=0,
# the LM head at every position); per arm the steps' times (the first two left out), their sum and median.
#   bash tools/settled_ab.sh [VAR=value ...] > gpurun_out/settled_ab.txt
set -e
ARGS="--seq-len 512 --micro-batch 32 --batch 512 --steps 24 --settled 0.5"
summarise() {
  python3 -c "
import ast, statistics, sys
t = [ast.literal_eval(l)['time/step'] for l in sys.stdin if l.startswith('{')][2:]
print(' '.join(f'{x:.3f}' for x in t)); print(f'sum {sum(t):.3f} s  median {statistics.median(t):.3f} s over {len(t)} steps')"
}
ARMS=("$@")
[ ${#ARMS[@]} -eq 0 ] && ARMS=(PRL_GATHER_SCORED=0)
for arm in default "${ARMS[@]}" default "${ARMS[@]}"; do
  echo "== $arm"
  if [ "$arm" = default ]; then
    timeout -k 10 240 python3 -u tools/ragged_rollouts_run.py $ARGS 2>&1 | summarise
  else
    env "$arm" timeout -k 10 240 python3 -u tools/ragged_rollouts_run.py $ARGS 2>&1 | summarise
  fi
done
