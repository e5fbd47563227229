"""Attention over a packed row as what it is: many short causal sequences.

The reference gives the packed row to flash-attn's variable-length kernel
through transformers (`attn_implementation = "flash_attention_2"` with position
ids that restart per sample, src/prime_rl/trainer/model.py:58-60, :251-254), so
attention costs the sum of the samples' squares.  flash-attn is a CUDA package;
PyTorch-ROCm's scaled_dot_product_attention takes the same row only with an
explicit [T, T] block-diagonal mask and then computes the full T x T product:
for the reverse_text recipe (16 samples of 128 tokens per row) 16 times the
work, 30 % of the whole trainer step.

Here the samples are laid side by side instead: [n_samples, heads, L_max, D],
plain causal SDPA over the batch dimension, no mask.  Equal-length samples (the
synthetic loader, fixed-length recipes) are a pure re-view of the row; ragged
ones are gathered into the padded layout and scattered back (padding sits at
the end of a sample, so under the causal rule no real query ever sees it).  The
layout is decided on the HOST from the micro-batch's position ids before they
are copied to the device, so nothing synchronises.
"""

from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import Any

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


@dataclass
class PackedPlan:
    total: int  # tokens in the row
    n: int  # samples
    lmax: int  # longest sample
    equal: bool  # every sample has lmax tokens
    gather: Tensor | None  # [n * lmax] row position feeding each padded slot (ragged rows only)
    scatter: Tensor | None  # [total] padded slot of each row position


def sample_starts(position_ids: Tensor) -> np.ndarray:
    """Start offsets of the sequences attention must keep apart: wherever the position does not continue the
    previous one (transformers' own rule for packed rows, masking_utils.find_packed_sequence_indices)."""
    pos = position_ids.reshape(-1).cpu().numpy()
    if pos.size == 0:
        return np.zeros(0, dtype=np.int64)
    brk = np.nonzero(np.diff(pos) != 1)[0] + 1
    return np.concatenate([np.zeros(1, dtype=np.int64), brk.astype(np.int64)])


def plan_from_position_ids(position_ids: Tensor, device: torch.device | str, max_work_ratio: float = 0.5):
    """A `PackedPlan` for one packed row [1, T] of HOST position ids, or None when the side-by-side layout would
    not at least halve the work (one long sample, or one sample much longer than the rest)."""
    if position_ids.ndim == 2 and position_ids.shape[0] != 1:
        return None
    starts = sample_starts(position_ids)
    total = int(position_ids.numel())
    n = int(starts.size)
    if n < 2:
        return None
    lengths = np.diff(np.concatenate([starts, [total]]))
    lmax = int(lengths.max())
    if n * lmax * lmax > max_work_ratio * total * total:
        return None
    if bool((lengths == lmax).all()):
        return PackedPlan(total, n, lmax, True, None, None)
    offs = np.arange(lmax, dtype=np.int64)[None, :]
    gather = starts[:, None] + np.minimum(offs, lengths[:, None] - 1)  # padding repeats the sample's last token
    seg = np.repeat(np.arange(n, dtype=np.int64), lengths)
    scatter = seg * lmax + (np.arange(total, dtype=np.int64) - starts[seg])
    g = torch.from_numpy(gather.reshape(-1)).to(device, non_blocking=True)
    s = torch.from_numpy(scatter).to(device, non_blocking=True)
    return PackedPlan(total, n, lmax, False, g, s)


_ACTIVE: list[PackedPlan | None] = [None]
_HF_SDPA: list[Any] = []


@contextlib.contextmanager
def use_plan(plan: PackedPlan | None):
    """The plan the attention of the forward passes inside this block follows."""
    prev = _ACTIVE[0]
    _ACTIVE[0] = plan
    try:
        yield
    finally:
        _ACTIVE[0] = prev


def packed_sdpa_attention(module: Any, query: Tensor, key: Tensor, value: Tensor, attention_mask: Tensor | None,
                          dropout: float = 0.0, scaling: float | None = None, **kwargs: Any):
    """Drop-in for transformers' `sdpa_attention_forward` (same arguments, same [batch, seq, heads, dim] result).
    Follows the active `PackedPlan`; without one, or for shapes the plan does not describe, the original runs."""
    plan = _ACTIVE[0]
    windowed = kwargs.get("sliding_window") is not None  # a sliding-window layer needs its own mask: not ours
    if plan is None or windowed or query.shape[0] != 1 or query.shape[2] != plan.total \
            or key.shape[2] != plan.total or dropout:
        return _HF_SDPA[0](module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, **kwargs)
    n, lmax = plan.n, plan.lmax
    gqa = key.shape[1] != query.shape[1]

    def side_by_side(x: Tensor) -> Tensor:  # [1, H, T, D] -> [n, H, lmax, D]
        x = x.squeeze(0)  # not x[0]: a select's backward is a zero fill plus a copy, a squeeze's is a view
        if not plan.equal:
            x = x.index_select(1, plan.gather)
        return x.unflatten(1, (n, lmax)).permute(1, 0, 2, 3)

    out = F.scaled_dot_product_attention(side_by_side(query), side_by_side(key), side_by_side(value), is_causal=True,
                                         scale=scaling, enable_gqa=gqa)
    out = out.permute(0, 2, 1, 3).reshape(n * lmax, query.shape[1], query.shape[3])  # [n * lmax, H, D]
    if not plan.equal:
        out = out.index_select(0, plan.scatter)
    return out.unsqueeze(0), None


def install() -> bool:
    """Put `packed_sdpa_attention` in place of transformers' "sdpa" attention for every model in the process.
    Returns False when this transformers has no attention registry."""
    try:
        from transformers import AttentionInterface
        from transformers.integrations.sdpa_attention import sdpa_attention_forward
    except Exception:
        return False
    if not _HF_SDPA:
        _HF_SDPA.append(sdpa_attention_forward)
    AttentionInterface.register("sdpa", packed_sdpa_attention)
    return True


def uninstall() -> None:
    if _HF_SDPA:
        from transformers import AttentionInterface

        AttentionInterface.register("sdpa", _HF_SDPA[0])
