#!/bin/bash
# Runs on the MI355X box: the two-pass loss path (PRL_FUSED_ROWS=0) and the fp32-logits path through bench.py, one
# line each: step time, the backward kernel's time and share of the HBM peak.
set -e -o pipefail
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],4), 'ms/step', d['roofline']['kernel'], round(d['roofline']['mean_ms']*1e3,1), 'us', round(d['roofline']['frac'],3), {k:round(v['mean_ms']*1e3,1) for k,v in d['kernels'].items()})"
}
PRL_FUSED_ROWS=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-trainer-extra --steps 20 --warmup 3 2>/dev/null | line "bf16 two-pass"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-trainer-extra --steps 20 --warmup 3 --logits-dtype f32 2>/dev/null | line "fp32 logits"
