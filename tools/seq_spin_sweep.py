"""How many sweeps does a workgroup of the one-pass sequence-mode kernel have to wait for its sample?  Rows left
to the second pass, and the time of `policy_loss` + backward, per spin limit, on ragged samples of 11-128
tokens and on samples of 200 tokens (T = 2048, V = 151936, nothing else on the GPU).

    python3 tools/seq_spin_sweep.py
"""
from __future__ import annotations

import os, random, statistics, sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import prime_rl_cse_amd as prl  # noqa: E402
from prime_rl_cse_amd import loss as L  # noqa: E402

t, v, dev = 2048, 151936, torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
bufs = []
for _ in range(3):
    x = torch.empty(1, t, v, dtype=torch.bfloat16, device=dev)
    for r in range(0, t, 512):
        x[0, r : r + 512] = (torch.randn(512, v, generator=g, device=dev) * 2.5).to(torch.bfloat16)
    bufs.append(x)
ids = torch.randint(0, v, (1, t), generator=g, device=dev)
adv = torch.randn(1, t, generator=g, device=dev)
old = torch.full((1, t), -12.0, device=dev)
mask = torch.ones(1, t, dtype=torch.bool, device=dev)
cfg = prl.LossConfig(ratio_type="sequence", ratio_length_norm=True)
unit = prl.unit_gradient(dev)


def positions(kind: str) -> torch.Tensor:
    if kind == "ragged":
        rng, parts, left = random.Random(7), [], t
        while left > 0:
            n = min(rng.randint(11, 128), left)
            parts.append(torch.arange(n))
            left -= n
        return torch.cat(parts).unsqueeze(0).to(dev)
    return torch.arange(200).repeat(t // 200 + 1)[:t].unsqueeze(0).to(dev)


count = [0]
for kind in ("ragged", "200"):
    pos = positions(kind)
    for spins in (4096, 1024, 256, 64, 32, 16, 8, 4, 0):
        os.environ["PRL_SEQ_ROWS_SPINS"] = str(spins)
        pend, times = [], []
        for rep in range(12):
            x = bufs[count[0] % 3].requires_grad_(True)
            count[0] += 1
            x.grad = None
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            loss, aux = prl.policy_loss(x, ids, pos, adv, old, mask, 1.0, cfg, 16.0)
            loss.backward(unit)
            b.record()
            torch.cuda.synchronize()
            if rep >= 2:
                times.append(a.elapsed_time(b) * 1e3)
                pend.append(int(aux["seq_pending"].sum()))
        print(f"samples {kind:7s} spin limit {spins:5d}: rows left to the second pass {min(pend):4d}..{max(pend):4d} of {t}, "
              f"{statistics.median(times):6.1f} us", flush=True)
