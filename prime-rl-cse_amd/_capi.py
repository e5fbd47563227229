"""ctypes binding of libprl_hip.so (the C ABI declared in include/prl_hip.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError
says so.  Every function takes raw device pointers and the caller's HIP stream,
so kernels stay ordered with whatever torch has already enqueued.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

import torch

PRL_F32, PRL_BF16 = 0, 1
RATIO_TOKEN, RATIO_SEQUENCE = 0, 1
ADVANTAGE_KIND = {"drgrpo": 0, "drgrpo-negclipped": 1, "rloo": 2, "opo": 3}

STAT_IDENTITY, STAT_EXP = 0, 1


class StatPiece(Structure):
    """prl_stat_piece of include/prl_hip.h."""

    _fields_ = [("x", c_void_p), ("mask", c_void_p), ("n", c_int64), ("key", c_int32), ("op", c_int32)]


LIB_PATH = Path(__file__).resolve().parent / "lib" / "libprl_hip.so"

# name -> (restype, argtypes), exactly the declarations of include/prl_hip.h
SIGNATURES = {
    "prl_last_error": (c_char_p, []),
    "prl_abi_version": (c_int, []),
    "prl_logprob_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "prl_logprob_entropy_fwd": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p,
         c_void_p, c_size_t, c_void_p],
    ),
    "prl_logprob_static_scratch_bytes": (c_size_t, []),
    "prl_logprob_entropy_fwd_static": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "prl_logprob_bwd": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
         c_void_p, c_int, c_int64, c_void_p],
    ),
    "prl_logprob_bwd_selected": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
         c_void_p, c_int, c_int64, c_void_p, c_void_p],
    ),
    "prl_policy_rows_seq_workspace_bytes": (c_size_t, [c_int64]),
    "prl_policy_rows_seq_fwd_bwd": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
         c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_int,
         c_void_p, c_size_t, c_void_p],
    ),
    "prl_segments_from_position_ids": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "prl_grpo_loss": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_float,
         c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "prl_policy_forward": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
         c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "prl_policy_tail": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "prl_policy_rows_supported": (c_int, [c_int, c_int64, c_int64, c_int64, c_int, c_int64]),
    "prl_policy_rows_fwd_bwd": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
         c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p],
    ),
    "prl_policy_rows_time_next": (c_int, [c_void_p, c_void_p]),
    "prl_policy_rows_gathered_fwd_bwd": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
         c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p],
    ),
    "prl_scale_rows_if_not_one": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "prl_swiglu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "prl_swiglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "prl_accumulate_chunk_elems": (c_int64, []),
    "prl_accumulate": (
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "prl_accumulate_first": (
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "prl_accumulate_squares": (
        c_int,
        [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "prl_attn_short_supported": (c_int, [c_int, c_int64, c_int64, c_int64, c_int64]),
    "prl_attn_short_fwd": (
        c_int,
        [c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p,
         POINTER(c_int64), c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_float, c_void_p],
    ),
    "prl_attn_short_bwd": (
        c_int,
        [c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p,
         POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p,
         POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, c_int32, c_int64, c_int32, c_int32, c_float,
         c_void_p],
    ),
    "prl_attn_long_supported": (c_int, [c_int, c_int64, c_int64, c_int64, c_int64]),
    "prl_attn_long_fwd": (
        c_int,
        [c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p,
         POINTER(c_int64), c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int32, c_int32, c_float, c_void_p],
    ),
    "prl_attn_long_bwd": (
        c_int,
        [c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p,
         POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p,
         POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, c_int32, c_int64, c_int64, c_int32, c_int32, c_float,
         c_void_p],
    ),
    "prl_swiglu_rows_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p]),
    "prl_swiglu_rows_bwd": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p],
    ),
    "prl_adamw_flat": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_float, c_float, c_float,
         c_float, c_float, c_int64, c_void_p],
    ),
    "prl_qk_norm_rope_supported": (c_int, [c_int, c_int64]),
    "prl_qk_norm_rope_bwd_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "prl_qk_norm_rope_fwd": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
         c_int64, c_int64, c_int64, c_float, c_void_p],
    ),
    "prl_qk_norm_rope_bwd": (
        c_int,
        [c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_int, c_int64, c_int64,
         c_int64, c_void_p],
    ),
    "prl_rope_supported": (c_int, [c_int, c_int64]),
    "prl_rope": (
        c_int,
        [c_void_p, c_void_p, POINTER(c_int64), c_int64, c_void_p, c_void_p, POINTER(c_int64), c_int64, c_void_p,
         c_void_p, c_int64, c_int64, c_int, c_int64, c_int64, c_int64, c_int, c_void_p],
    ),
    "prl_rmsnorm_supported": (c_int, [c_int, c_int64]),
    "prl_rmsnorm_bwd_workspace_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "prl_rmsnorm_fwd": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int64, c_float, c_int64,
         c_int64, c_void_p],
    ),
    "prl_rmsnorm_bwd": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64,
         c_int64, c_void_p, c_size_t, c_int64, c_int64, c_void_p],
    ),
    "prl_add_rmsnorm_fwd": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int,
         c_int64, c_int64, c_float, c_void_p],
    ),
    "prl_add_rmsnorm_bwd": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
         c_int, c_int64, c_int64, c_void_p, c_size_t, c_void_p],
    ),
    "prl_sum_parts": (c_int, [c_void_p, c_int, c_int64, c_int32, c_int64, c_void_p, c_void_p]),
    "prl_lm_head_supported": (c_int, [c_int, c_int64, c_int64, c_int64]),
    "prl_lm_head_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "prl_lm_head_fwd": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p],
    ),
    "prl_lm_head_bwd_input": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_size_t,
         c_void_p],
    ),
    "prl_lm_head_bwd_weight": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_size_t,
         c_void_p],
    ),
    "prl_graph_node_counts": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    "prl_group_advantage": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "prl_stats_workspace_bytes": (c_size_t, [c_int]),
    "prl_stats_layout": (c_int, [c_int, POINTER(c_int64)]),
    "prl_stats_begin": (c_int, [c_void_p, c_int, c_void_p]),
    "prl_stats_pass": (c_int, [c_int, POINTER(StatPiece), c_int, c_int, c_void_p, c_void_p]),
    "prl_stats_finish": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "prl_masked_stats_workspace_bytes": (c_size_t, []),
    "prl_masked_stats": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
}

_lib: ctypes.CDLL | None = None


def load() -> ctypes.CDLL:
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("PRL_HIP_LIB", LIB_PATH))
        if not path.exists():
            raise RuntimeError(
                f"libprl_hip.so not found at {path}: build it with `make -C prime-rl-cse_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path"
            )
        lib = ctypes.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here means header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().prl_last_error().decode() or "no message"
        raise RuntimeError(f"{what} failed with code {code}: {msg}")


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return PRL_F32
    if t.dtype == torch.bfloat16:
        return PRL_BF16
    raise RuntimeError(f"logits must be float32 or bfloat16, got {t.dtype}")


def require_device(*tensors: torch.Tensor) -> torch.device:
    """All tensors on one HIP device; that device is returned."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "this path runs on a HIP device only (got a tensor on "
                f"{t.device}); move the micro-batch to the GPU first, there is no CPU fallback"
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} and {t.device}")
    return dev


class _OnDevice:
    """`with on_device(dev):` makes `dev` current for the launches inside; when it already is (the usual case:
    one process per GPU) entering and leaving cost nothing."""

    __slots__ = ("ctx",)

    def __init__(self, device: torch.device):
        self.ctx = None if device.index is None or torch.cuda.current_device() == device.index else \
            torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def on_device(device: torch.device) -> _OnDevice:
    return _OnDevice(device)


def stream_ptr(device: torch.device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t: torch.Tensor | None) -> c_void_p:
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


# --------------------------------------------------------------------------- optional per-launch timing


class KernelTimer:
    """HIP events around named launches, recorded on the stream the launch goes
    to (torch's current stream).  Meant for benchmarks: install one with
    `set_timer`, run, synchronise, read `summary()`."""

    def __init__(self, only: set[str] | None = None, every: int = 1, kernel_stamps: int = 0,
                 device: torch.device | None = None):
        """`kernel_stamps` > 0: launches of prl_policy_rows_fwd_bwd are timed by the kernel's own start and end on
        the GPU (`prl_policy_rows_time_next`) instead of by an event recorded on either side of the call; that
        many pairs of events are made ready here, on `device`, outside any timed region (an event exists only
        once it has been recorded)."""
        self.events: dict[str, list[tuple[torch.cuda.Event, torch.cuda.Event]]] = {}
        self.only = only  # names to time; None = every named launch
        self.every = max(int(every), 1)  # time one launch in `every` (an event pair drains the queue around it)
        self._seen: dict[str, int] = {}
        self._stamps: list[tuple[torch.cuda.Event, torch.cuda.Event]] = []
        if kernel_stamps > 0:
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
            st = torch.cuda.current_stream(dev)
            for _ in range(kernel_stamps):
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                pair[0].record(st)
                pair[1].record(st)
                self._stamps.append(pair)
            st.synchronize()

    def span(self, name: str, device: torch.device):
        if self.only is not None and name not in self.only:
            return _no_span
        n = self._seen.get(name, 0)
        self._seen[name] = n + 1
        if n % self.every:
            return _no_span
        if name == "prl_policy_rows_fwd_bwd" and self._stamps:
            return _StampSpan(self, name, self._stamps.pop())
        return _Span(self, name, device)

    def summary(self) -> dict[str, dict[str, float]]:
        out = {}
        for name, pairs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = {"launches": len(ms), "mean_ms": sum(ms) / len(ms), "min_ms": min(ms), "max_ms": max(ms)}
        return out


class _Span:
    def __init__(self, timer: KernelTimer, name: str, device: torch.device):
        self.timer, self.name, self.device = timer, name, device

    def __enter__(self):
        self.start = torch.cuda.Event(enable_timing=True)
        self.stop = torch.cuda.Event(enable_timing=True)
        self.start.record(torch.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        self.stop.record(torch.cuda.current_stream(self.device))
        self.timer.events.setdefault(self.name, []).append((self.start, self.stop))
        return False


class _StampSpan:
    """The row kernel's own start and end: the two events are handed to the library, which attaches them to the
    launch the enclosed call makes."""

    def __init__(self, timer: KernelTimer, name: str, pair: tuple[torch.cuda.Event, torch.cuda.Event]):
        self.timer, self.name, self.pair = timer, name, pair

    def __enter__(self):
        check(load().prl_policy_rows_time_next(c_void_p(self.pair[0].cuda_event), c_void_p(self.pair[1].cuda_event)),
              "prl_policy_rows_time_next")

    def __exit__(self, *exc):
        load().prl_policy_rows_time_next(None, None)  # (already taken by the launch; cleared if the call failed)
        self.timer.events.setdefault(self.name, []).append(self.pair)
        return False


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_timer: KernelTimer | None = None
_no_span = _NoSpan()


def set_timer(timer: KernelTimer | None) -> None:
    global _timer
    _timer = timer


def timed(name: str, device: torch.device):
    return _timer.span(name, device) if _timer is not None else _no_span
