#!/bin/bash
# Run one gpurun call as soon as the pool has a free box: retries ONLY while gpurun
# reports "no box or slot free" (exit code 3), never after the command itself ran.
# usage: tools/gpu_when_free.sh <timeout-seconds> '<command>'
limit="$1"; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$limit" -- "$@"
  rc=$?
  if [ "$rc" -ne 3 ]; then exit "$rc"; fi
  sleep 90
done
exit 3
