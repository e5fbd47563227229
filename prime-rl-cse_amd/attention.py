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

from .batch import host_to_device


@dataclass
class PackedPlan:
    total: int  # tokens in the row
    n: int  # samples
    lmax: int  # longest sample
    equal: bool  # every sample has lmax tokens
    gather: Tensor | None  # [n * lmax] row position feeding each padded slot (ragged rows only)
    scatter: Tensor | None  # [total] padded slot of each row position
    cu: Tensor | None = None  # int32 [n + 1] on the device: first token of every sample, then `total`
    side_by_side: bool = True  # whether the padded layout at least halves the work of the masked row


def sample_starts(position_ids: Tensor) -> np.ndarray:
    """Start offsets of the sequences attention must keep apart: wherever the position does not continue the
    previous one (transformers' own rule for packed rows, masking_utils.find_packed_sequence_indices)."""
    pos = position_ids.reshape(-1).cpu().numpy()
    if pos.size == 0:
        return np.zeros(0, dtype=np.int64)
    brk = np.nonzero(np.diff(pos) != 1)[0] + 1
    return np.concatenate([np.zeros(1, dtype=np.int64), brk.astype(np.int64)])


def plan_from_position_ids(position_ids: Tensor, device: torch.device | str, max_work_ratio: float = 0.5):
    """A `PackedPlan` for one packed row [1, T] of HOST position ids (None for an empty row or a batch of rows):
    the sample table every variable-length kernel takes and, where padding to the longest sample at least halves
    the work of the masked row, the index tables of the side-by-side layout."""
    if position_ids.ndim == 2 and position_ids.shape[0] != 1:
        return None
    starts = sample_starts(position_ids)
    total = int(position_ids.numel())
    n = int(starts.size)
    if n == 0:
        return None
    lengths = np.diff(np.concatenate([starts, [total]]))
    lmax = int(lengths.max())
    cu = host_to_device(torch.from_numpy(np.concatenate([starts, [total]]).astype(np.int32)), device)
    side_by_side = n >= 2 and n * lmax * lmax <= max_work_ratio * total * total
    if not side_by_side:
        return PackedPlan(total, n, lmax, False, None, None, cu, False)
    if bool((lengths == lmax).all()):
        return PackedPlan(total, n, lmax, True, None, None, cu)
    offs = np.arange(lmax, dtype=np.int64)[None, :]
    gather = starts[:, None] + np.minimum(offs, lengths[:, None] - 1)  # padding repeats the sample's last token
    seg = np.repeat(np.arange(n, dtype=np.int64), lengths)
    scatter = seg * lmax + (np.arange(total, dtype=np.int64) - starts[seg])
    g = host_to_device(torch.from_numpy(gather.reshape(-1)), device)
    sc = host_to_device(torch.from_numpy(scatter), device)
    return PackedPlan(total, n, lmax, False, g, sc, cu)


SHORT_MAX_LEN = 128  # longest sample the short-sample kernels take (and the only head dimension: 128)


def _short_kernel_enabled() -> bool:
    """PRL_ATTN_SHORT=0 sends short samples through SDPA too (for A/B measurements)."""
    import os

    return os.environ.get("PRL_ATTN_SHORT", "1") != "0"


def _strides2(x: Tensor):
    from . import _capi as C

    return (C.c_int64 * 2)(x.stride(2), x.stride(1))  # [1, heads, tokens, d] view: token stride, head stride


def _kernel_ready(x: Tensor) -> Tensor:
    """[1, heads, tokens, d] with d contiguous, 16-byte aligned, strides in multiples of 8 elements; else a copy."""
    if x.stride(3) != 1 or x.data_ptr() % 16 or x.stride(1) % 8 or x.stride(2) % 8:
        return x.contiguous()
    return x


class _AttnShort(torch.autograd.Function):
    """q [1, Hq, T, 128], k / v [1, Hkv, T, 128] (any token / head strides) -> out [1, T, Hq, 128]."""

    @staticmethod
    def forward(ctx, q: Tensor, k: Tensor, v: Tensor, cu: Tensor, n: int, scale: float):
        from . import _capi as C

        dev = C.require_device(q, k, v, cu)
        lib = C.load()
        q, k, v = _kernel_ready(q), _kernel_ready(k), _kernel_ready(v)
        _, hq, t, d = q.shape
        hkv = k.shape[1]
        out = torch.empty((1, t, hq, d), dtype=q.dtype, device=dev)
        lse2 = torch.empty((hq, t), dtype=torch.float32, device=dev)
        ov = out.transpose(1, 2)
        with C.on_device(dev), C.timed("prl_attn_short_fwd", dev):
            C.check(
                lib.prl_attn_short_fwd(C.ptr(q), _strides2(q), C.ptr(k), _strides2(k), C.ptr(v), _strides2(v),
                                       C.ptr(out), _strides2(ov), C.ptr(lse2), C.ptr(cu), n, t, hq, hkv, scale,
                                       C.stream_ptr(dev)),
                "prl_attn_short_fwd",
            )
        ctx.save_for_backward(q, k, v, out, lse2, cu)
        ctx.n, ctx.scale = n, scale
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        from . import _capi as C

        q, k, v, out, lse2, cu = ctx.saved_tensors
        dev = q.device
        lib = C.load()
        _, hq, t, d = q.shape
        hkv = k.shape[1]
        do = _kernel_ready(dout.transpose(1, 2))
        ov = out.transpose(1, 2)
        dq, dk, dv = _stacked_grads(t, hq, hkv, d, q.dtype, dev)
        delta = torch.empty((hq, t), dtype=torch.float32, device=dev)
        with C.on_device(dev), C.timed("prl_attn_short_bwd", dev):
            C.check(
                lib.prl_attn_short_bwd(C.ptr(q), _strides2(q), C.ptr(k), _strides2(k), C.ptr(v), _strides2(v),
                                       C.ptr(ov), _strides2(ov), C.ptr(do), _strides2(do), C.ptr(lse2), C.ptr(delta),
                                       C.ptr(dq), _strides2(dq), C.ptr(dk), _strides2(dk), C.ptr(dv), _strides2(dv),
                                       C.ptr(cu), ctx.n, t, hq, hkv, ctx.scale, C.stream_ptr(dev)),
                "prl_attn_short_bwd",
            )
        return dq, dk, dv, None, None, None


def _stacked_grads(t: int, hq: int, hkv: int, d: int, dtype: torch.dtype, dev: torch.device):
    """dq, dk, dv as [1, heads, t, d] views of ONE [t, (hq + 2 hkv) d] matrix, query heads first: the layout of the
    gradient of a stacked q/k/v projection's output, so the block in front of the attention can take the three
    as that matrix without copying (`fused._QKVNormRope.backward`); to anyone else they are three tensors."""
    buf = torch.empty((1, t, hq + 2 * hkv, d), dtype=dtype, device=dev)
    return (buf[:, :, :hq].transpose(1, 2), buf[:, :, hq : hq + hkv].transpose(1, 2),
            buf[:, :, hq + hkv :].transpose(1, 2))


class _AttnLong(torch.autograd.Function):
    """The same for samples of any length (`prl_attn_long_fwd` / `_bwd`: flash-style tiles with a running
    soft-max); `lmax` is the longest sample of the table."""

    @staticmethod
    def forward(ctx, q: Tensor, k: Tensor, v: Tensor, cu: Tensor, n: int, lmax: int, scale: float):
        from . import _capi as C

        dev = C.require_device(q, k, v, cu)
        lib = C.load()
        q, k, v = _kernel_ready(q), _kernel_ready(k), _kernel_ready(v)
        _, hq, t, d = q.shape
        hkv = k.shape[1]
        out = torch.empty((1, t, hq, d), dtype=q.dtype, device=dev)
        lse2 = torch.empty((hq, t), dtype=torch.float32, device=dev)
        ov = out.transpose(1, 2)
        with C.on_device(dev), C.timed("prl_attn_long_fwd", dev):
            C.check(
                lib.prl_attn_long_fwd(C.ptr(q), _strides2(q), C.ptr(k), _strides2(k), C.ptr(v), _strides2(v),
                                      C.ptr(out), _strides2(ov), C.ptr(lse2), C.ptr(cu), n, lmax, t, hq, hkv, scale,
                                      C.stream_ptr(dev)),
                "prl_attn_long_fwd",
            )
        ctx.save_for_backward(q, k, v, out, lse2, cu)
        ctx.n, ctx.lmax, ctx.scale = n, lmax, scale
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        from . import _capi as C

        q, k, v, out, lse2, cu = ctx.saved_tensors
        dev = q.device
        lib = C.load()
        _, hq, t, d = q.shape
        hkv = k.shape[1]
        do = _kernel_ready(dout.transpose(1, 2))
        ov = out.transpose(1, 2)
        dq, dk, dv = _stacked_grads(t, hq, hkv, d, q.dtype, dev)
        delta = torch.empty((hq, t), dtype=torch.float32, device=dev)
        with C.on_device(dev), C.timed("prl_attn_long_bwd", dev):
            C.check(
                lib.prl_attn_long_bwd(C.ptr(q), _strides2(q), C.ptr(k), _strides2(k), C.ptr(v), _strides2(v),
                                      C.ptr(ov), _strides2(ov), C.ptr(do), _strides2(do), C.ptr(lse2), C.ptr(delta),
                                      C.ptr(dq), _strides2(dq), C.ptr(dk), _strides2(dk), C.ptr(dv), _strides2(dv),
                                      C.ptr(cu), ctx.n, ctx.lmax, t, hq, hkv, ctx.scale, C.stream_ptr(dev)),
                "prl_attn_long_bwd",
            )
        return dq, dk, dv, None, None, None, None


def long_attention_supported(query: Tensor, key: Tensor, plan: "PackedPlan") -> bool:
    """Samples of any length through this package's tiled kernels (PRL_ATTN_LONG=0: leave them to torch)."""
    import os

    if plan.cu is None or not query.is_cuda or query.dtype != torch.bfloat16 \
            or os.environ.get("PRL_ATTN_LONG", "1") == "0":
        return False
    from . import _capi as C

    return bool(C.load().prl_attn_long_supported(C.dtype_code(query), query.shape[3], plan.lmax, query.shape[1],
                                                 key.shape[1]))


def long_attention(query: Tensor, key: Tensor, value: Tensor, plan: "PackedPlan", scaling: float | None = None):
    """Causal attention within every sample of a packed row, samples of any length; [1, heads, T, 128] in,
    [1, T, heads, 128] out."""
    scale = float(scaling) if scaling is not None else float(query.shape[3]) ** -0.5
    return _AttnLong.apply(query, key, value, plan.cu, plan.n, plan.lmax, scale)


def _varlen_attn():
    """PyTorch's variable-length attention, or None (older torch, or PRL_ATTN_VARLEN=0 for A/B measurements)."""
    import os

    if os.environ.get("PRL_ATTN_VARLEN", "1") == "0":
        return None
    try:
        from torch.nn.attention.varlen import varlen_attn
    except Exception:
        return None
    return varlen_attn


def varlen_attention_supported(query: Tensor, plan: "PackedPlan", scaling: float | None) -> bool:
    if plan.cu is None or not query.is_cuda or query.dtype not in (torch.bfloat16, torch.float16):
        return False
    if scaling is not None and abs(float(scaling) - float(query.shape[3]) ** -0.5) > 1e-12:
        return False  # the kernel's scale is fixed at 1 / sqrt(head dimension)
    return _varlen_attn() is not None


def varlen_attention(query: Tensor, key: Tensor, value: Tensor, plan: "PackedPlan") -> Tensor:
    """[1, heads, T, d] in, [1, T, heads, d] out, through torch's varlen_attn on the packed row."""
    rows = lambda x: x.squeeze(0).transpose(0, 1)  # [T, heads, d]
    out = _varlen_attn()(rows(query), rows(key), rows(value), plan.cu, plan.cu, plan.lmax, plan.lmax, is_causal=True)
    return out.unsqueeze(0)


def short_attention_supported(query: Tensor, key: Tensor, plan: "PackedPlan") -> bool:
    if plan.cu is None or plan.lmax > SHORT_MAX_LEN or not query.is_cuda or query.dtype != torch.bfloat16 \
            or not _short_kernel_enabled():
        return False
    from . import _capi as C

    return bool(C.load().prl_attn_short_supported(C.dtype_code(query), query.shape[3], plan.lmax, query.shape[1],
                                                  key.shape[1]))


def short_attention(query: Tensor, key: Tensor, value: Tensor, plan: "PackedPlan", scaling: float | None = None):
    """Causal attention within every sample of a packed row of short samples; [1, heads, T, 128] in,
    [1, T, heads, 128] out (the layout transformers' attention functions return)."""
    scale = float(scaling) if scaling is not None else float(query.shape[3]) ** -0.5
    return _AttnShort.apply(query, key, value, plan.cu, plan.n, scale)


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


def maskless(plan: PackedPlan | None, model: Any) -> bool:
    """Whether a forward of `model` under `plan` never looks at the attention mask, so that the model need not
    build one: transformers makes a [T, T] boolean mask per forward (two passes over 268 MB at 16 K tokens, 0.4 ms)
    that every path below except the masked row ignores.  True only where all layers are plain causal attention
    through the registered "sdpa" function, without dropout, and either the plan lays the samples side by side
    (then the short-sample kernels, `varlen_attn` or the padded layout take every layer) or the model computes in
    a half-precision type that `varlen_attn` takes (then that does, whatever the samples' lengths).  Such a
    forward passes `attention_mask={"full_attention": None}`; a layer that turns out to need the mask after all
    raises."""
    cfg = getattr(model, "config", None)
    kinds = getattr(cfg, "layer_types", None)
    if not (plan is not None and _HF_SDPA and kinds and all(k == "full_attention" for k in kinds)
            and getattr(cfg, "_attn_implementation", None) == "sdpa"
            and not float(getattr(cfg, "attention_dropout", 0.0) or 0.0)):
        return False
    if plan.side_by_side:
        return True
    first = next(iter(model.parameters()), None)
    return bool(first is not None and first.is_cuda and first.dtype in (torch.bfloat16, torch.float16)
                and plan.cu is not None and _varlen_attn() is not None)


def _masked_row(module: Any, query: Tensor, key: Tensor, value: Tensor, attention_mask: Tensor | None, plan: Any,
                **kwargs: Any):
    """transformers' own attention over the whole row under its block-diagonal mask."""
    if attention_mask is None and plan is not None and plan.n > 1:
        raise RuntimeError("packed-row attention: this layer needs the block-diagonal mask, which was not built "
                           "(attention.maskless said no layer would)")
    return _HF_SDPA[0](module, query, key, value, attention_mask, **kwargs)


def packed_sdpa_attention(module: Any, query: Tensor, key: Tensor, value: Tensor, attention_mask: Tensor | None,
                          dropout: float = 0.0, scaling: float | None = None, **kwargs: Any):
    """Drop-in for transformers' `sdpa_attention_forward` (same arguments, same [batch, seq, heads, dim] result).
    Follows the active `PackedPlan`; without one, or for shapes the plan does not describe, the original runs."""
    plan = _ACTIVE[0]
    windowed = kwargs.get("sliding_window") is not None  # a sliding-window layer needs its own mask: not ours
    if plan is None or windowed or query.shape[0] != 1 or query.shape[2] != plan.total \
            or key.shape[2] != plan.total or dropout:
        return _masked_row(module, query, key, value, attention_mask, plan, dropout=dropout, scaling=scaling,
                           **kwargs)
    if short_attention_supported(query, key, plan):
        return short_attention(query, key, value, plan, scaling), None
    if long_attention_supported(query, key, plan):
        return long_attention(query, key, value, plan, scaling), None
    if varlen_attention_supported(query, plan, scaling):
        return varlen_attention(query, key, value, plan), None
    if not plan.side_by_side:
        return _masked_row(module, query, key, value, attention_mask, plan, dropout=dropout, scaling=scaling,
                           **kwargs)
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
