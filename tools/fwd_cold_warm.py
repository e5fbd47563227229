"""The two-pass forward kernel (`k_logprob_fwd`, one 622 MB read of bf16 logits) by what the caches hold when it runs:
the benchmark's situation (eight static logits buffers in turn: nothing of the buffer is resident), the trainer's
(the logits were written just before, by the LM head's product), and the same buffer read again and again.

    python3 tools/fwd_cold_warm.py
"""
from __future__ import annotations

import statistics
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import prime_rl_cse_amd as prl  # noqa: E402

t, v, dev = 2048, 151936, torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
bufs = []
for _ in range(8):
    x = torch.empty(1, t, v, dtype=torch.bfloat16, device=dev)
    for r in range(0, t, 512):
        x[0, r : r + 512] = (torch.randn(512, v, generator=g, device=dev) * 2.5).to(torch.bfloat16)
    bufs.append(x)
ids = torch.randint(0, v, (1, t), generator=g, device=dev)


def timed(prepare, pick, rounds=24):
    out = []
    for i in range(rounds):
        x = pick(i)
        prepare(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        prl.selective_log_softmax(prl.shift_logits(x), ids)
        b.record()
        torch.cuda.synchronize()
        if i >= 4:
            out.append(a.elapsed_time(b) * 1e3)
    return statistics.median(out), min(out)


with torch.no_grad():
    cases = {
        "eight static buffers in turn (the benchmark)": (lambda x: None, lambda i: bufs[i % 8]),
        "buffer rewritten just before (the trainer: the LM head's product)": (lambda x: x.copy_(bufs[7]), lambda i: bufs[i % 4]),
        "the same buffer again and again": (lambda x: None, lambda i: bufs[0]),
    }
    for name, (prepare, pick) in cases.items():
        med, best = timed(prepare, pick)
        print(f"{name:68s} median {med:6.1f} us  min {best:6.1f} us  {t * v * 2 / med / 1e6:5.2f} TB/s", flush=True)
