"""Rollouts -> packed micro-batches per trainer rank, the layout the kernels
consume (src/prime_rl/orchestrator/batch.py:21-175).  Host code: it runs on the
orchestrator, once per step, over a few hundred short integer lists."""

from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Any

import numpy as np
import torch
from torch import Tensor

from .data import MicroBatch


@dataclass
class Rollout:
    """One sampled completion (src/prime_rl/orchestrator/buffer.py:19-29)."""

    problem_id: int
    prompt_tokens: list[int]
    prompt_mask: list[int]
    completion_tokens: list[int]
    completion_mask: list[int]
    completion_logprobs: list[float]
    is_truncated: bool
    reward: float
    advantage: float


BatchSample = dict  # input_ids, position_ids, loss_mask, advantages, logprobs: 1-D tensors of one length
_KEYS = ("input_ids", "advantages", "loss_mask", "position_ids", "logprobs")


def prepare_sample(rollout: Rollout, seq_len: int, tokenizer: Any = None) -> BatchSample:
    """Per-token tensors of one rollout: prompt then completion; prompt tokens
    carry log-probability 0; every token carries the rollout's advantage
    (src/prime_rl/orchestrator/batch.py:21-62)."""
    n_prompt, n_comp = len(rollout.prompt_tokens), len(rollout.completion_tokens)
    n = n_prompt + n_comp
    if n > seq_len:
        raise ValueError(f"Number of tokens {n} is greater than sequence length {seq_len}. This should not happen.")
    ids = np.empty(n, dtype=np.int64)
    ids[:n_prompt] = rollout.prompt_tokens
    ids[n_prompt:] = rollout.completion_tokens
    mask = np.empty(n, dtype=np.bool_)
    mask[:n_prompt] = np.asarray(rollout.prompt_mask, dtype=np.int64) != 0
    mask[n_prompt:] = np.asarray(rollout.completion_mask, dtype=np.int64) != 0
    lp = np.zeros(n, dtype=np.float32)
    lp[n_prompt:] = np.asarray(rollout.completion_logprobs, dtype=np.float32)
    assert len(rollout.prompt_mask) == n_prompt and len(rollout.completion_mask) == n_comp
    assert len(rollout.completion_logprobs) == n_comp
    return {
        "input_ids": torch.from_numpy(ids),
        "advantages": torch.full((n,), float(rollout.advantage), dtype=torch.float32),
        "loss_mask": torch.from_numpy(mask),
        "position_ids": torch.arange(n, dtype=torch.int64),
        "logprobs": torch.from_numpy(lp),
    }


def packed_samples_into_micro_bs(samples: list[BatchSample], max_seq_len: int) -> list[list[BatchSample]]:
    """First-fit-decreasing: samples in stable descending order of length, each
    into the first bin that still has room (src/prime_rl/orchestrator/batch.py:76-102).
    The fill of every bin is kept, so the cost is O(samples x bins)."""
    lengths = [int(s["input_ids"].shape[0]) for s in samples]
    order = sorted(range(len(samples)), key=lengths.__getitem__, reverse=True)
    bins: list[list[BatchSample]] = []
    fill: list[int] = []
    for i in order:
        n = lengths[i]
        for b, used in enumerate(fill):
            if used + n <= max_seq_len:
                bins[b].append(samples[i])
                fill[b] = used + n
                break
        else:
            bins.append([samples[i]])
            fill.append(n)
    return bins


def prepare_micro_batch_packing(samples: list[BatchSample], max_seq_len: int, temperature: float) -> MicroBatch:
    """The samples of one bin back to back as a [1, T] row, T = the sum of their
    lengths: rows are NOT padded to the bin capacity
    (src/prime_rl/orchestrator/batch.py:105-120)."""
    assert sum(int(s["input_ids"].shape[0]) for s in samples) <= max_seq_len, (
        "Total tokens of samples is greater than max sequence length"
    )
    row: MicroBatch = {k: torch.cat([s[k] for s in samples], dim=0).unsqueeze(0) for k in _KEYS}
    row["temperature"] = temperature
    return row


def prepare_batch(
    rollouts: list[Rollout],
    temperature: float,
    tokenizer: Any,
    batch_size: int,
    micro_batch_size: int,
    seq_len: int,
    num_train_workers: int,
) -> list[list[MicroBatch]]:
    """One list of packed rows per trainer rank (src/prime_rl/orchestrator/batch.py:123-175).

    Every rank must run the same number of rows (the sharded model issues its
    collectives per row), so with W > 1 ranks `W - n % W` filler rows are
    appended: copies of row 0 with zero advantages and an all-False mask, which
    add nothing to the loss.  That is a whole extra W when n already divides W,
    as in the reference.  Rows are dealt to the ranks in contiguous runs."""
    capacity = seq_len * micro_batch_size
    samples = [prepare_sample(r, capacity, tokenizer) for r in rollouts]
    rows = [prepare_micro_batch_packing(b, capacity, temperature) for b in packed_samples_into_micro_bs(samples, capacity)]
    w = num_train_workers
    n_fill = w - len(rows) % w
    if w > 1 and n_fill > 0:
        filler = copy.deepcopy(rows[0])
        filler["advantages"] = torch.zeros_like(filler["advantages"])
        filler["loss_mask"] = torch.zeros_like(filler["loss_mask"], dtype=torch.bool)
        rows.extend([filler] * n_fill)
    assert len(rows) % w == 0, "Number of micro batches is not divisible by number of data ranks"
    per = len(rows) // w
    return [rows[r * per : (r + 1) * per] for r in range(w)]


import os as _os

_PINNED = _os.environ.get("PRL_PINNED_STAGING", "1") != "0"


def host_to_device(x: Any, device: torch.device | str, dtype: torch.dtype | None = None) -> Tensor:
    """A host tensor (or list of numbers) on `device` without draining the stream: through pinned memory.

    A copy out of pageable host memory waits for everything already enqueued on the stream before it starts, even
    when asked not to block: in the middle of a step that is a bubble the length of the host's own lag (0.5 ms
    behind a backward pass).  A pinned source makes the copy one more stream-ordered operation.  The pinned
    buffer comes from torch's caching host allocator, which holds it until the copy has run."""
    t = x if isinstance(x, Tensor) else torch.tensor(x, dtype=dtype)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device.type == "cpu" and torch.device(device).type == "cuda" and _PINNED:
        if not t.is_pinned():
            t = t.pin_memory()
        return t.to(device, non_blocking=True)
    return t.to(device, non_blocking=True)


def to_device(micro_batch: MicroBatch, device: torch.device | str, non_blocking: bool = True) -> MicroBatch:
    """The host-to-device copies of src/prime_rl/trainer/rl/train.py:275-279 for one row (staged through pinned
    memory, `host_to_device`; `non_blocking=False` gives plain blocking copies)."""
    if not non_blocking:
        return {k: (v.to(device) if isinstance(v, Tensor) else v) for k, v in micro_batch.items()}
    return {k: (host_to_device(v, device) if isinstance(v, Tensor) else v) for k, v in micro_batch.items()}
