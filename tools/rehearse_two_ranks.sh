#!/bin/bash
# Runs on the one-GPU box: the N = 2 bench line with two ranks sharing the device and exchanging over gloo
# through host memory (a rehearsal of the control flow; the RCCL runs need one GPU per rank).
set -e -o pipefail
mkdir -p gpurun_out
PRL_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 "$@" \
  > gpurun_out/two_ranks.json 2> gpurun_out/two_ranks.err
tail -c 1500 gpurun_out/two_ranks.json
