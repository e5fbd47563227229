#!/bin/bash
# Runs on the one-GPU box: the N = 2 bench line with two ranks sharing the device and exchanging over gloo
# through host memory (a rehearsal of the control flow; the RCCL runs need one GPU per rank).  bench.py starts
# its own ranks.
set -e -o pipefail
mkdir -p gpurun_out
PRL_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 2 "$@" \
  > gpurun_out/two_ranks.json 2> gpurun_out/two_ranks.err
tail -c 1500 gpurun_out/two_ranks.json
