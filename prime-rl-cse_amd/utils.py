"""Packed-row bookkeeping and logged statistics on the device
(counterparts of src/prime_rl/trainer/utils.py:36-69 and :195-256)."""

from __future__ import annotations

from collections import defaultdict

import torch
import torch.distributed as dist
from torch import Tensor

from . import _capi as C
from .loss import segments_from_position_ids

STAT_NAMES = ("mean", "median", "std", "min", "max")


def get_response_lengths(position_ids: Tensor) -> list[int]:
    """Lengths of the samples concatenated in one packed row
    (src/prime_rl/trainer/utils.py:36-69), bit for bit.

    The boundaries are found by one kernel; a single device-to-host copy then
    fetches the table (the reference pays about three copies per token)."""
    _, seg_start, n_seg = segments_from_position_ids(position_ids)
    table = torch.cat([n_seg.to(torch.int64), seg_start]).cpu()
    n = int(table[0])
    starts = table[1 : n + 2]
    return (starts[1:] - starts[:-1]).tolist()


def masked_stats(x: Tensor, mask: Tensor | None = None) -> tuple[Tensor, Tensor]:
    """(fp32 [5] = mean, median, std, min, max; int64 [1] = how many entries)
    of the selected entries of `x`, computed on the device with no
    synchronisation.  std is the unbiased one and the median of an even count
    is the lower middle value, as torch's are."""
    dev = C.require_device(x, mask)
    lib = C.load()
    xf = x.detach().reshape(-1).to(torch.float32).contiguous()
    m8 = None
    if mask is not None:
        m = mask.reshape(-1)
        if m.numel() != xf.numel():
            raise ValueError("masked_stats: mask and x differ in size")
        m8 = m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8).contiguous()
    out = torch.empty(5, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    ws_bytes = lib.prl_masked_stats_workspace_bytes()
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        C.check(
            lib.prl_masked_stats(
                C.ptr(xf), C.ptr(m8), xf.numel(), C.ptr(out), C.ptr(count), C.ptr(ws), ws.numel() * 8,
                C.stream_ptr(dev),
            ),
            "prl_masked_stats",
        )
    return out, count


def flexible_all_gather(tensor: Tensor) -> Tensor:
    """All-gather a 1-D device tensor whose length differs per rank
    (src/prime_rl/trainer/utils.py:195-225): the sizes travel first, then one
    padded gather; the result is the concatenation in rank order."""
    assert tensor.ndim == 1, "Can only flexibly all-gather 1D tensors"
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tensor
    world = dist.get_world_size()
    sizes = torch.zeros(world, dtype=torch.int64, device=tensor.device)
    dist.all_gather_into_tensor(sizes, torch.tensor([tensor.numel()], dtype=torch.int64, device=tensor.device))
    sizes = sizes.tolist()
    width = max(sizes)
    padded = torch.zeros(width, dtype=tensor.dtype, device=tensor.device)
    padded[: tensor.numel()] = tensor
    gathered = torch.empty(world * width, dtype=tensor.dtype, device=tensor.device)
    dist.all_gather_into_tensor(gathered, padded)
    return torch.cat([gathered[r * width : r * width + n] for r, n in enumerate(sizes)])


class Tensors(defaultdict):
    """Accumulates per-micro-batch vectors ON THE DEVICE and reduces each key to
    mean / median / std / min / max over all micro-batches and ranks
    (src/prime_rl/trainer/utils.py:228-256).

    Unlike the reference nothing is copied to the host while the step runs:
    `compute_stats` makes one device-to-host copy of 5 floats per key at the end."""

    def __init__(self):
        super().__init__(list)
        self._masks: dict[str, list[Tensor | None]] = defaultdict(list)

    def add(self, key: str, values: Tensor, mask: Tensor | None = None) -> None:
        """Record `values[mask]` for `key` without forming it: the selection is
        applied inside the statistics kernel, so no host synchronisation (which a
        boolean index needs to size its result) happens while the step runs."""
        self[key].append(values.detach())
        self._masks[key].append(mask)

    def compute_stats(self) -> dict[str, float]:
        keys = list(self.keys())
        outs = []
        multi = dist.is_initialized() and dist.get_world_size() > 1
        for key in keys:
            parts = self.pop(key)
            masks = self._masks.pop(key, [])
            masks = masks + [None] * (len(parts) - len(masks))  # entries appended the reference's way carry no mask
            vec = torch.cat([t.reshape(-1).float() for t in parts], dim=0)
            sel = None
            if any(m is not None for m in masks):
                sel = torch.cat(
                    [
                        (torch.ones(t.numel(), dtype=torch.bool, device=t.device) if m is None else m.reshape(-1).bool())
                        for t, m in zip(parts, masks)
                    ]
                )
            if multi:
                vec = flexible_all_gather(vec if sel is None else vec[sel])
                sel = None
            out, _ = masked_stats(vec, sel)
            outs.append(out)
        if not outs:
            return {}
        host = torch.stack(outs).cpu()
        metrics: dict[str, float] = {}
        for key, row in zip(keys, host.tolist()):
            for name, val in zip(STAT_NAMES, row):
                metrics[f"{key}/{name}"] = val
        return metrics
