"""Optimizers and learning-rate schedules of the trainer
(counterparts of src/prime_rl/trainer/optim.py:9-57 and src/prime_rl/trainer/scheduler.py:7-105).

The schedules are written as one closed-form factor of the step instead of a
chain of torch schedulers; `tests/test_optim.py` checks that they trace the same
learning rates as the reference's composition of LinearLR / CosineAnnealingLR /
SequentialLR."""

from __future__ import annotations

import math
from typing import Any, Callable

import torch
from torch import nn
from torch.optim.lr_scheduler import LambdaLR, LRScheduler


def setup_optimizer(config: Any, model: nn.Module, fused: bool | None = None) -> torch.optim.Optimizer:
    """The optimizer of src/prime_rl/trainer/optim.py:9-38 over `model.parameters()`.  `fused=True` asks torch for
    its one-kernel AdamW (same update rule; worth it over one large flat parameter)."""
    if config.type == "sgd":
        return torch.optim.SGD(model.parameters(), lr=config.lr, weight_decay=config.weight_decay,
                               momentum=config.momentum, nesterov=config.nesterov)
    if config.type == "adamw":
        return torch.optim.AdamW(model.parameters(), lr=config.lr, weight_decay=config.weight_decay,
                                 betas=(config.betas1, config.betas2), **({} if fused is None else {"fused": fused}))
    if config.type == "muon":
        raise NotImplementedError("the Muon optimizer comes from the `dion` package, which is not installed here")
    raise ValueError(f"Invalid optimizer type: {config.type}")


def _floor_factor(lr: float, min_lr: float) -> float:
    return min_lr / lr if min_lr > 0 else 1e-8


def linear_factor(max_steps: int | None, warmup_steps: int, decay_steps: int, lr: float, min_lr: float) -> Callable[[int], float]:
    """Warm-up, plateau, linear decay (src/prime_rl/trainer/scheduler.py:12-47)."""
    assert warmup_steps > 0 or decay_steps > 0, "Either warmup steps or decay steps must be specified for a linear scheduler"
    floor = _floor_factor(lr, min_lr)
    decay_start = None
    if decay_steps > 0:
        assert max_steps is not None, "max_steps must be specified when specifying decay_steps"
        decay_start = max_steps - decay_steps
        assert decay_start >= warmup_steps

    def factor(step: int) -> float:
        if step < warmup_steps:
            return floor + (1.0 - floor) * step / warmup_steps
        if decay_start is None or step < decay_start:
            return 1.0
        span = decay_steps - 1  # the reference decays over decay_steps - 1 intervals
        k = min(step - decay_start, span)
        return 1.0 + (floor - 1.0) * (k / span if span > 0 else 1.0)

    return factor


def cosine_factor(max_steps: int | None, warmup_steps: int, lr: float, min_lr: float) -> Callable[[int], float]:
    """Warm-up, then half a cosine down to min_lr (src/prime_rl/trainer/scheduler.py:50-75)."""
    assert max_steps is not None, "max_steps must be specified when specifying decay_steps"
    floor = _floor_factor(lr, min_lr)
    t_max = max_steps - warmup_steps

    def factor(step: int) -> float:
        if step < warmup_steps:
            return floor + (1.0 - floor) * step / warmup_steps
        k = step - warmup_steps
        return (min_lr + (lr - min_lr) * (1.0 + math.cos(math.pi * k / t_max)) / 2.0) / lr

    return factor


def setup_scheduler(optimizer: torch.optim.Optimizer, config: Any, max_steps: int | None, lr: float) -> LRScheduler:
    if config.type == "constant":
        return LambdaLR(optimizer, lambda step: 1.0)
    if config.type == "linear":
        return LambdaLR(optimizer, linear_factor(max_steps, config.warmup_steps, config.decay_steps, lr, config.min_lr))
    if config.type == "cosine":
        return LambdaLR(optimizer, cosine_factor(max_steps, config.warmup_steps, lr, config.min_lr))
    raise ValueError(f"Invalid scheduler type: {config.type}")
