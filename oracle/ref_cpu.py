"""CPU oracle for the policy-update hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, with torch on the CPU, the arithmetic of the reference
trainer between "the model returned logits" and "the gradient with respect to
those logits exists", plus the group advantage and the packing of rollouts into
micro-batches.  It is the checker the HIP path is compared against; it is never
the thing that ships or the thing that is measured.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.

Parity pin: every function here is checked bit for bit (integers) or to exact
float equality against the reference's own functions by ``oracle/make_golden.py``,
which loads the reference files in the build container and writes the vectors
kept under ``tests/golden/``.  The reference's own unit tests hold no numeric
vectors for this path (``tests/unit/train/rl/test_loss.py:10-48`` asserts
shapes only); the only known answers are the advantage docstrings
(``src/prime_rl/orchestrator/advantage.py:13-16, 27-30``), which
``tests/test_oracle.py`` checks as well.

Reference files followed (relative to the reference root):
  src/prime_rl/trainer/rl/loss.py:13-124
  src/prime_rl/trainer/utils.py:36-69
  src/prime_rl/orchestrator/advantage.py:9-103
  src/prime_rl/orchestrator/batch.py:21-175
  src/prime_rl/trainer/rl/train.py:259-333   (call order of the epilogue)
  src/prime_rl/trainer/rl/data.py:39-54      (synthetic micro-batch)
"""

from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Any, Sequence

import numpy as np
import torch
from torch import Tensor


# --------------------------------------------------------------------------- config


@dataclass
class LossConfig:
    """The three loss knobs (src/prime_rl/trainer/rl/config.py:19-27)."""

    ratio_type: str = "token"  # "token" | "sequence"
    ratio_length_norm: bool = False
    clip_ratio: float = 8.0


# --------------------------------------------------------------------------- logits side


def shift_logits(logits: Tensor) -> Tensor:
    """Row t of the result is row t-1 of the input; row 0 is all zeros and the
    last input row is dropped (src/prime_rl/trainer/rl/loss.py:51-58)."""
    b, _, v = logits.shape
    head = logits.new_zeros(b, 1, v)
    return torch.cat((head, logits[:, :-1, :]), dim=1)


def selective_log_softmax(logits: Tensor, index: Tensor) -> Tensor:
    """log softmax(logits)[index] per position (src/prime_rl/trainer/rl/loss.py:13-39).

    fp32/fp64 take ``x[index] - logsumexp(x)`` one batch row at a time; lower
    precisions take a full log_softmax per batch row and then gather."""
    picker = index.unsqueeze(-1)
    if logits.dtype in (torch.float32, torch.float64):
        chosen = logits.gather(-1, picker).squeeze(-1)
        lse = torch.stack([torch.logsumexp(row, dim=-1) for row in logits])
        return chosen - lse
    rows = []
    for row, lab in zip(logits, index):
        rows.append(torch.log_softmax(row, dim=-1).gather(-1, lab.unsqueeze(-1)).squeeze(-1))
    return torch.stack(rows)


def compute_entropy(shifted_logits: Tensor) -> Tensor:
    """H = logsumexp(x) - sum(softmax(x) * x) (src/prime_rl/trainer/rl/loss.py:43-47)."""
    p = torch.softmax(shifted_logits, dim=-1)
    return torch.logsumexp(shifted_logits, dim=-1) - (p * shifted_logits).sum(dim=-1)


# --------------------------------------------------------------------------- packed-row bookkeeping


def get_response_lengths(position_ids: Tensor) -> list[int]:
    """Lengths of the samples packed into one row (src/prime_rl/trainer/utils.py:36-69).

    Index i (1 <= i) starts a new sample iff pos[i] == 0, pos[i-1] != 0, i+1 is
    inside the row and pos[i+1] == 1.  Every other zero (padding, or a
    one-token sample) stays with the sample before it."""
    pos = position_ids.flatten().cpu().numpy()
    n = pos.shape[0]
    if n < 3:
        return [int(n)]
    mid = np.flatnonzero((pos[1:-1] == 0) & (pos[:-2] != 0) & (pos[2:] == 1)) + 1
    starts = np.concatenate(([0], mid, [n]))
    return [int(x) for x in np.diff(starts)]


# --------------------------------------------------------------------------- loss


def compute_loss(
    logprobs: Sequence[Tensor],
    old_logprobs: Sequence[Tensor],
    advantages: Sequence[Tensor],
    loss_mask: Sequence[Tensor],
    loss_config: Any,
    loss_scale: float,
) -> tuple[Tensor, dict[str, Tensor]]:
    """Ratio-clip policy loss over the samples of one packed row
    (src/prime_rl/trainer/rl/loss.py:61-124).

    token mode:    ratio_t = exp(lp_t - old_t), one per token
    sequence mode: one ratio per sample from the masked sum of (lp - old),
                   optionally divided by the masked count, capped at log 10;
                   the sample's loss is divided by its masked count
    both:          clipped = min(ratio, clip_ratio); loss = sum_masked(-clipped * adv);
                   is_clipped = ratio > clip_ratio; total / max(loss_scale, 1)."""
    per_sequence = loss_config.ratio_type == "sequence"
    total = 0
    ratios, clipped_ratios, clip_flags = [], [], []
    for lp, old, adv, mask in zip(logprobs, old_logprobs, advantages, loss_mask):
        log_ratio = lp - old
        if per_sequence:
            s = log_ratio[mask].sum()
            if loss_config.ratio_length_norm:
                s = s / torch.clamp_min(mask.sum(), 1)
            log_ratio = torch.clamp(s.unsqueeze(0), max=10.0)
        ratio = torch.exp(log_ratio)
        capped = torch.clamp(ratio, max=loss_config.clip_ratio)
        term = -capped * adv
        seg_loss = term[mask].sum()
        if per_sequence:
            seg_loss = seg_loss / torch.clamp_min(mask.sum(), 1)
        total = total + seg_loss
        ratios.append(ratio)
        clipped_ratios.append(capped)
        clip_flags.append((ratio > loss_config.clip_ratio).float())
    return total / max(loss_scale, 1), {
        "importance_ratio": torch.cat(ratios),
        "clipped_importance_ratio": torch.cat(clipped_ratios),
        "is_clipped": torch.cat(clip_flags),
    }


# --------------------------------------------------------------------------- the epilogue as the trainer runs it


def policy_epilogue(
    logits: Tensor,
    input_ids: Tensor,
    position_ids: Tensor,
    advantages: Tensor,
    old_logprobs: Tensor,
    loss_mask: Tensor,
    temperature: float,
    loss_config: Any,
    loss_scale: float,
    want_grad: bool = True,
) -> dict[str, Any]:
    """One micro-batch of the trainer's hot loop from the model's logits to the
    gradient with respect to them (src/prime_rl/trainer/rl/train.py:285-310):
    float copy, shift, divide by the temperature, per-token logprob, sample
    lengths, loss, entropy without grad, backward.

    ``logits`` is the model output [1, T, V] in its own dtype (bf16 in the
    trainer, because FSDP runs the model in bf16)."""
    leaf = logits.detach().clone().requires_grad_(want_grad)
    x = leaf.float().contiguous()
    shifted = shift_logits(x) / temperature
    lp = selective_log_softmax(shifted, input_ids)
    lengths = get_response_lengths(position_ids)
    loss, aux = compute_loss(
        lp.squeeze().split(lengths),
        old_logprobs.squeeze().split(lengths),
        advantages.squeeze().split(lengths),
        loss_mask.squeeze().split(lengths),
        loss_config,
        loss_scale,
    )
    with torch.no_grad():
        ent = compute_entropy(shifted)
    out = {
        "logprobs": lp.detach(),
        "entropy": ent,
        "lengths": lengths,
        "loss": loss.detach(),
        "importance_ratio": aux["importance_ratio"].detach(),
        "clipped_importance_ratio": aux["clipped_importance_ratio"].detach(),
        "is_clipped": aux["is_clipped"].detach(),
    }
    del x, shifted
    if want_grad:
        if loss.requires_grad:
            loss.backward()
            out["dlogits"] = leaf.grad
        else:  # e.g. a row with no sample at all
            out["dlogits"] = torch.zeros_like(leaf)
    return out


# --------------------------------------------------------------------------- advantages


def _adv_drgrpo(r: Tensor, _: Tensor) -> Tensor:
    return r - r.mean()


def _adv_negclipped(r: Tensor, _: Tensor) -> Tensor:
    return torch.maximum(r - r.mean(), torch.zeros_like(r))


def _adv_rloo(r: Tensor, _: Tensor) -> Tensor:
    g = r.shape[0]
    return (r - r.mean()) * g / (g - 1)


def _adv_opo(r: Tensor, lengths: Tensor) -> Tensor:
    w = lengths.to(r.dtype)
    return r - (r * w).sum() / w.sum()


ADVANTAGE_KINDS = {
    "drgrpo": _adv_drgrpo,
    "drgrpo-negclipped": _adv_negclipped,
    "rloo": _adv_rloo,
    "opo": _adv_opo,
}


def compute_advantages(
    rewards: list[float], completion_lengths: list[int], samples_per_problem: int, advantage_type: str
) -> list[float]:
    """Group-relative advantages, one group of ``samples_per_problem``
    consecutive rewards at a time (src/prime_rl/orchestrator/advantage.py:71-103)."""
    assert len(rewards) % samples_per_problem == 0
    fn = ADVANTAGE_KINDS[advantage_type]
    out: list[float] = []
    for lo in range(0, len(rewards), samples_per_problem):
        r = torch.tensor(rewards[lo : lo + samples_per_problem])
        n = torch.tensor(completion_lengths[lo : lo + samples_per_problem])
        out.extend(fn(r, n).tolist())
    assert len(out) == len(rewards)
    return out


# --------------------------------------------------------------------------- packing rollouts into micro-batches


@dataclass
class Rollout:
    """The rollout fields the packer reads (src/prime_rl/orchestrator/buffer.py:19-29)."""

    problem_id: int
    prompt_tokens: list[int]
    prompt_mask: list[int]
    completion_tokens: list[int]
    completion_mask: list[int]
    completion_logprobs: list[float]
    is_truncated: bool
    reward: float
    advantage: float


TOKEN_KEYS = ("input_ids", "advantages", "loss_mask", "position_ids", "logprobs")


def prepare_sample(rollout: Any, seq_len: int) -> dict[str, Tensor]:
    """One rollout as per-token tensors (src/prime_rl/orchestrator/batch.py:21-62)."""
    ids = torch.tensor(list(rollout.prompt_tokens) + list(rollout.completion_tokens)).long()
    mask = torch.tensor(list(rollout.prompt_mask) + list(rollout.completion_mask)).long().bool()
    lp = torch.cat(
        [torch.zeros(len(rollout.prompt_tokens)), torch.tensor(rollout.completion_logprobs)]
    ).float()
    n = ids.shape[0]
    if n > seq_len:
        raise ValueError(f"Number of tokens {n} is greater than sequence length {seq_len}. This should not happen.")
    return {
        "input_ids": ids,
        "advantages": torch.tensor(rollout.advantage).repeat(n).float(),
        "loss_mask": mask,
        "position_ids": torch.arange(n).long(),
        "logprobs": lp,
    }


def first_fit_decreasing(lengths: Sequence[int], capacity: int) -> list[list[int]]:
    """Bins of sample indices: samples in stable descending length order, each
    into the first bin it still fits (src/prime_rl/orchestrator/batch.py:76-102)."""
    order = sorted(range(len(lengths)), key=lambda i: lengths[i], reverse=True)
    bins: list[list[int]] = []
    fill: list[int] = []
    for i in order:
        for b in range(len(bins)):
            if fill[b] + lengths[i] <= capacity:
                bins[b].append(i)
                fill[b] += lengths[i]
                break
        else:
            bins.append([i])
            fill.append(lengths[i])
    return bins


def prepare_batch(
    rollouts: list[Any],
    temperature: float,
    batch_size: int,
    micro_batch_size: int,
    seq_len: int,
    num_train_workers: int,
) -> list[list[dict[str, Any]]]:
    """Micro-batches per trainer rank (src/prime_rl/orchestrator/batch.py:123-175).

    Rows are not padded to the bin capacity.  With more than one trainer rank,
    ``W - n % W`` filler rows are appended (a whole extra W when n already
    divides): copies of row 0 with zero advantages and an all-False mask.  Rows
    are then dealt to the ranks in contiguous runs."""
    capacity = seq_len * micro_batch_size
    samples = [prepare_sample(r, capacity) for r in rollouts]
    bins = first_fit_decreasing([s["input_ids"].shape[0] for s in samples], capacity)
    rows: list[dict[str, Any]] = []
    for members in bins:
        row: dict[str, Any] = {k: torch.cat([samples[i][k] for i in members]).unsqueeze(0) for k in TOKEN_KEYS}
        row["temperature"] = temperature
        rows.append(row)
    w = num_train_workers
    n_fill = w - len(rows) % w
    if w > 1 and n_fill > 0:
        filler = copy.deepcopy(rows[0])
        filler["advantages"] = torch.zeros_like(filler["advantages"])
        filler["loss_mask"] = torch.zeros_like(filler["loss_mask"], dtype=torch.bool)
        rows.extend([filler] * n_fill)
    assert len(rows) % w == 0
    per = len(rows) // w
    return [rows[r * per : (r + 1) * per] for r in range(w)]


# --------------------------------------------------------------------------- synthetic data


def fake_micro_batch(micro_batch_size: int, seq_len: int, generator: torch.Generator | None = None) -> dict[str, Any]:
    """The reference's debug micro-batch (src/prime_rl/trainer/rl/data.py:39-54),
    with an explicit generator so that runs repeat."""
    t = micro_batch_size * seq_len
    return {
        "input_ids": torch.randint(0, 100, (1, t), generator=generator),
        "position_ids": torch.cat([torch.arange(seq_len)] * micro_batch_size).unsqueeze(0),
        "advantages": torch.randn(t, generator=generator).unsqueeze(0),
        "logprobs": torch.randn(t, generator=generator).unsqueeze(0),
        "temperature": 1.0,
        "loss_mask": torch.ones(t, dtype=torch.bool).unsqueeze(0),
    }


# --------------------------------------------------------------------------- statistics


def tensor_stats(x: Tensor) -> dict[str, float]:
    """mean / median / std / min / max as the trainer logs them
    (src/prime_rl/trainer/utils.py:247-251): std is the unbiased one and the
    median of an even count is the lower of the two middle values."""
    return {
        "mean": x.mean().item(),
        "median": torch.median(x).item(),
        "std": x.std().item(),
        "min": x.min().item(),
        "max": x.max().item(),
    }
