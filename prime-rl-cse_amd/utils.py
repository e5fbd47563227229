"""Packed-row bookkeeping and logged statistics on the device
(counterparts of src/prime_rl/trainer/utils.py:36-69 and :195-256)."""

from __future__ import annotations

import contextlib

from collections import defaultdict
from typing import Any

import torch
import torch.distributed as dist
from torch import Tensor

from . import _capi as C
from .loss import _u8, segments_from_position_ids

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
        if mask.numel() != xf.numel():
            raise ValueError("masked_stats: mask and x differ in size")
        m8 = _u8(mask)
    out = torch.empty(5, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    ws_bytes = lib.prl_masked_stats_workspace_bytes()
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=dev)
    with C.on_device(dev):
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
    padded gather; the result is the concatenation in rank order.  Kept for
    callers that want the values themselves; the statistics below do not
    gather."""
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


def _all_reduce(t: Tensor, op) -> None:
    """In-place all-reduce of a device tensor.  RCCL ("nccl") reduces it where it
    is; any other backend (gloo in tests) gets a synchronous host copy, because
    gloo collectives on device tensors are not reliable when ranks share a GPU."""
    if dist.get_backend() == "nccl":
        dist.all_reduce(t, op=op)
        return
    host = t.cpu()
    dist.all_reduce(host, op=op)
    t.copy_(host)


class Tensors(defaultdict):
    """Accumulates per-micro-batch vectors ON THE DEVICE and reduces each key to
    mean / median / std / min / max over all micro-batches and all ranks
    (src/prime_rl/trainer/utils.py:228-256).

    Nothing is copied to the host while the step runs and nothing is gathered:
    every key is a list of device pieces, all keys are reduced together by four
    histogram passes, and with several ranks the small scratch (about 4 KiB per
    key) is all-reduced between the passes, which yields the exact statistics of
    the union of all ranks' values.  `compute_stats` ends with one
    device-to-host copy of 5 floats per key.

    `tensors[key].append(x)` works as in the reference; `add` additionally takes
    a selection mask and an `exp` flag so that neither `x[mask]` (a host
    synchronisation) nor `torch.exp(x)` (an extra pass) has to be formed."""

    # the keys the trainer logs (src/prime_rl/trainer/rl/train.py:313-333, :241-245), in the order every rank
    # reduces them when there are several ranks
    SHARED_KEYS = ("probs", "old_probs", "entropy", "loss", "importance_ratio", "clipped_importance_ratio",
                   "is_clipped", "recomputed_logprob_error")

    def __init__(self):
        super().__init__(list)
        self._extra: dict[int, tuple[Tensor | None, int]] = {}

    def add(self, key: str, values: Tensor, mask: Tensor | None = None, exp: bool = False) -> None:
        v = values.detach()
        self[key].append(v)
        self._extra[id(v)] = (mask, C.STAT_EXP if exp else C.STAT_IDENTITY)

    def compute_stats(self) -> dict[str, float]:
        """Reduce every key and return {"<key>/<stat>": float}; blocks until the
        numbers are on the host."""
        return self.compute_stats_async().result()

    def compute_stats_async(self) -> "StatsFuture":
        """Enqueue the reduction and the device-to-host copy of its result (into
        pinned memory) and return at once.  `result()` on the returned object
        waits for that copy only, so a trainer can log step N while step N + 1 is
        already running.

        With several ranks the launches and the seven small all-reduces between
        the passes go to a stream of their own that first waits for everything
        enqueued so far, so that the next step's kernels do not queue behind
        collectives whose cost is latency.  On one GPU there is nothing to hide
        but 45 us of statistics and the cross-stream waits cost more than that
        (2.01 vs 1.97 ms per bench step), so a single rank stays on the current
        stream.  PRL_STATS_STREAM=1 / =0 forces either.  The pieces are
        referenced by the returned object until `result()` has been called."""
        multi = dist.is_initialized() and dist.get_world_size() > 1
        local = list(self.keys())
        # Between ranks the scratch of ALL keys is all-reduced as one block per region, so every rank must reduce
        # the same keys in the same order, also a rank that logged fewer of them this step (no row with a
        # gradient, no recomputed log-probabilities): the trainer's keys in a fixed order, then whatever else
        # was logged, sorted (such extra keys must then be logged on every rank).
        keys = local if not multi else \
            list(self.SHARED_KEYS) + sorted(k for k in local if k not in self.SHARED_KEYS)
        if not keys or (not multi and not local):
            return StatsFuture([], None, None)
        lib = C.load()
        keep: list[Tensor] = []  # the pieces' storage must outlive the launches
        pieces = []
        dev = None
        for k, key in enumerate(keys):
            for part in (self.pop(key) if key in self else ()):
                mask, op = self._extra.pop(id(part), (None, C.STAT_IDENTITY))
                dev = C.require_device(part, mask)
                x = part.reshape(-1).to(torch.float32).contiguous()
                m8 = None if mask is None else _u8(mask)
                if m8 is not None and m8.numel() != x.numel():
                    raise ValueError(f"Tensors: mask and values of {key!r} differ in size")
                keep += [x, m8]
                pieces.append(C.StatPiece(x.data_ptr(), 0 if m8 is None else m8.data_ptr(), x.numel(), k, op))
        self._extra.clear()
        n_keys, n_pieces = len(keys), len(pieces)
        table = (C.StatPiece * max(n_pieces, 1))(*pieces)
        off = (C.c_int64 * 5)()
        C.check(lib.prl_stats_layout(n_keys, off), "prl_stats_layout")
        if dev is None:  # a rank that logged nothing at all still takes part in the other ranks' all-reduces
            dev = torch.device("cuda", torch.cuda.current_device())
        main = torch.cuda.current_stream(dev)
        side = _stats_stream(dev, multi)
        if side is not None:
            side.wait_stream(main)  # the pieces were produced on the current stream
        with C.on_device(dev), (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            ws = torch.empty(off[4] // 8, dtype=torch.int64, device=dev)
            out = torch.empty((n_keys, 5), dtype=torch.float32, device=dev)
            st = C.stream_ptr(dev)
            with C.timed("prl_stats", dev):
                C.check(lib.prl_stats_begin(C.ptr(ws), n_keys, st), "prl_stats_begin")
                for level in range(4):
                    C.check(lib.prl_stats_pass(level, table, n_pieces, n_keys, C.ptr(ws), st), "prl_stats_pass")
                    if multi:
                        i32 = ws.view(torch.int32)
                        h0 = off[3] // 4 + level * n_keys * 256
                        _all_reduce(i32[h0 : h0 + n_keys * 256], dist.ReduceOp.SUM)
                        if level == 0:
                            _all_reduce(ws[off[0] // 8 : off[0] // 8 + 2 * n_keys], dist.ReduceOp.SUM)
                            _all_reduce(ws.view(torch.float64)[off[1] // 8 : off[1] // 8 + 2 * n_keys],
                                        dist.ReduceOp.SUM)
                            _all_reduce(i32[off[2] // 4 : off[2] // 4 + 2 * n_keys], dist.ReduceOp.MAX)
                counts = torch.empty(n_keys, dtype=torch.int64, device=dev) if multi else None
                C.check(lib.prl_stats_finish(C.ptr(ws), n_keys, C.ptr(out), C.ptr(counts), st), "prl_stats_finish")
            host = torch.empty((n_keys, 5), dtype=torch.float32, pin_memory=True)
            host.copy_(out, non_blocking=True)
            host_counts = None
            if multi:
                host_counts = torch.empty(n_keys, dtype=torch.int64, pin_memory=True)
                host_counts.copy_(counts, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
        # the launches hold raw pointers into the pieces, `ws` and `out`: the future keeps all of them alive until
        # the copy-out has been seen to finish
        return StatsFuture(keys, host, done, keep + [ws, out, counts], host_counts, set(local))


_STATS_STREAMS: dict[int, Any] = {}


def _stats_stream(dev: torch.device, multi: bool):
    """The per-device stream the statistics run on (default priority), or None for "the current stream":
    their own stream when ranks exchange between the passes, unless PRL_STATS_STREAM says otherwise."""
    import os

    want = os.environ.get("PRL_STATS_STREAM", "auto")
    if want == "0" or (want != "1" and not multi):
        return None
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _STATS_STREAMS.get(idx)
    if s is None:
        s = _STATS_STREAMS[idx] = torch.cuda.Stream(device=dev)
    return s


class StatsFuture:
    """The pending result of `Tensors.compute_stats_async`."""

    def __init__(self, keys: list[str], host: Tensor | None, done: "torch.cuda.Event | None", keep: Any = None,
                 counts: Tensor | None = None, local: set[str] | None = None):
        self._keys, self._host, self._done, self._keep = keys, host, done, keep
        self._counts, self._local = counts, local

    def result(self) -> dict[str, float]:
        if self._host is None:
            return {}
        self._done.synchronize()  # the one host wait of the step's statistics
        self._keep = None
        metrics: dict[str, float] = {}
        counts = None if self._counts is None else self._counts.tolist()
        for i, (key, row) in enumerate(zip(self._keys, self._host.tolist())):
            if counts is not None and counts[i] == 0 and key not in self._local:
                continue  # one of the shared keys that no rank logged this step
            for name, val in zip(STAT_NAMES, row):
                metrics[f"{key}/{name}"] = val
        return metrics
