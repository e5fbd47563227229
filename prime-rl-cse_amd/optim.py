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


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW's update over ONE flat fp32 parameter on the GPU (the master shard of
    `replica.FlatState`), as one HIP launch that also applies the clipping factor, writes the compute copy of the
    new weights and clears the gradient (`prl_adamw_flat`).  Learning-rate schedulers drive it like any optimizer."""

    def __init__(self, params: Any, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1 or len(self.param_groups[0]["params"]) != 1:
            raise ValueError("FlatAdamW takes exactly one (flat) parameter")

    @torch.no_grad()
    def step_flat(self, grad: torch.Tensor, grad_scale: torch.Tensor | None = None,
                  compute_copy: torch.Tensor | None = None, zero_grad: bool = False) -> None:
        """One step with `grad` (fp32, same size as the parameter) times `grad_scale` (a 1-element device tensor,
        or None); `compute_copy` receives the new weights in its own dtype; `zero_grad` clears `grad`."""
        from . import _capi as C

        group = self.param_groups[0]
        p = group["params"][0]
        dev = C.require_device(p, grad, grad_scale, compute_copy)
        if p.dtype != torch.float32 or grad.dtype != torch.float32 or grad.numel() != p.numel() \
                or not (p.is_contiguous() and grad.is_contiguous()):
            raise ValueError("FlatAdamW: parameter and gradient must be contiguous fp32 tensors of one size")
        if compute_copy is not None and (compute_copy.numel() != p.numel() or not compute_copy.is_contiguous()):
            raise ValueError("FlatAdamW: the compute copy must be contiguous and as long as the parameter")
        state = self.state[p]
        if not state:
            state["step"] = 0
            state["exp_avg"] = torch.zeros_like(p)
            state["exp_avg_sq"] = torch.zeros_like(p)
        state["step"] += 1
        self._opt_called = True  # what torch's schedulers look at to tell a step has been taken
        b1, b2 = group["betas"]
        with C.on_device(dev), C.timed("prl_adamw_flat", dev):
            C.check(
                C.load().prl_adamw_flat(C.ptr(p), C.ptr(grad), C.ptr(state["exp_avg"]), C.ptr(state["exp_avg_sq"]),
                                        C.ptr(compute_copy), 0 if compute_copy is None else C.dtype_code(compute_copy),
                                        p.numel(), C.ptr(grad_scale), int(zero_grad), float(group["lr"]), float(b1),
                                        float(b2), float(group["eps"]), float(group["weight_decay"]), state["step"],
                                        C.stream_ptr(dev)),
                "prl_adamw_flat",
            )

    @torch.no_grad()
    def step(self, closure: Any = None) -> None:  # the plain Optimizer protocol: the parameter's own .grad
        p = self.param_groups[0]["params"][0]
        if p.grad is not None:
            self.step_flat(p.grad)


def setup_optimizer(config: Any, model: nn.Module, fused: bool | None = None, flat: bool = False) -> torch.optim.Optimizer:
    """The optimizer of src/prime_rl/trainer/optim.py:9-38 over `model.parameters()`.  `fused` asks torch for its
    one-kernel AdamW (same update rule) or for the per-operation one; left at None it is the one-kernel form
    wherever the parameters are on a GPU: over FSDP2's 311 sharded parameters of a 0.6B model the per-operation
    form costs the step 23 ms of small launches (135 -> 112 ms).  `flat=True` (one flat fp32 parameter on the
    GPU) gives `FlatAdamW`."""
    if config.type == "adamw" and flat:
        return FlatAdamW(model.parameters(), lr=config.lr, weight_decay=config.weight_decay,
                         betas=(config.betas1, config.betas2))
    if config.type == "sgd":
        return torch.optim.SGD(model.parameters(), lr=config.lr, weight_decay=config.weight_decay,
                               momentum=config.momentum, nesterov=config.nesterov)
    if config.type == "adamw":
        if fused is None:
            first = next(iter(model.parameters()), None)
            fused = bool(first is not None and first.is_cuda)
        return torch.optim.AdamW(model.parameters(), lr=config.lr, weight_decay=config.weight_decay,
                                 betas=(config.betas1, config.betas2), fused=fused)
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
