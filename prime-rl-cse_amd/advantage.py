"""Group-relative advantages on the device under the reference's signature
(src/prime_rl/orchestrator/advantage.py:60-103)."""

from __future__ import annotations

from typing import Literal

import torch
from torch import Tensor

from . import _capi as C

AdvantageType = Literal["drgrpo", "drgrpo-negclipped", "rloo", "opo"]


def group_advantage(rewards: Tensor, completion_lengths: Tensor | None, samples_per_problem: int, advantage_type: str) -> Tensor:
    """fp32 [B] advantages from device tensors; no host synchronisation."""
    if advantage_type not in C.ADVANTAGE_KIND:
        raise KeyError(advantage_type)
    dev = C.require_device(rewards, completion_lengths)
    lib = C.load()
    r = rewards.reshape(-1).to(torch.float32).contiguous()
    n = None if completion_lengths is None else completion_lengths.reshape(-1).to(torch.int64).contiguous()
    out = torch.empty_like(r)
    with C.on_device(dev):
        C.check(
            lib.prl_group_advantage(
                C.ptr(r), C.ptr(n), r.numel(), samples_per_problem, C.ADVANTAGE_KIND[advantage_type], C.ptr(out),
                C.stream_ptr(dev),
            ),
            "prl_group_advantage",
        )
    return out


def compute_advantages(
    rewards: list[float],
    completion_lengths: list[int],
    samples_per_problem: int,
    advantage_type: AdvantageType,
    device: torch.device | str = "cuda",
) -> list[float]:
    """Advantages from a flattened list of rewards, the first
    `samples_per_problem` belonging to the first problem and so on
    (src/prime_rl/orchestrator/advantage.py:71-103)."""
    assert len(rewards) % samples_per_problem == 0
    r = torch.tensor(rewards, dtype=torch.float32).to(device)
    n = torch.tensor(completion_lengths, dtype=torch.int64).to(device)
    out = group_advantage(r, n, samples_per_problem, advantage_type).cpu().tolist()
    assert len(out) == len(rewards)
    return out
