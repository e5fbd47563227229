"""Policy-update loss functions on HIP kernels, under the reference's names.

The callables keep the signatures of src/prime_rl/trainer/rl/loss.py
(`shift_logits` :51, `selective_log_softmax` :13, `compute_entropy` :43,
`compute_loss` :61), so a training loop written against the reference reads the
same.  Underneath, the vocabulary-wide work is one streaming pass over the
logits in `prl_logprob_entropy_fwd` and one in `prl_logprob_bwd`; nothing
shifted, scaled or soft-maxed is ever written to memory.

`policy_loss` is the same computation as one call with no host synchronisation
at all: sample boundaries, log-probabilities, entropy, loss and the gradient
seed stay on the device.
"""

from __future__ import annotations

from typing import Any, Sequence

import torch
from torch import Tensor

from . import _capi as C

# --------------------------------------------------------------------------- shifted logits as a view


class ShiftedLogits:
    """The model's logits [batch, seq, vocab] together with "read one position
    back" and "divide by the temperature", kept as facts instead of copies.

    `shift_logits(x)` returns one; dividing it by a temperature returns another.
    `selective_log_softmax` and `compute_entropy` accept it and apply both facts
    while streaming the original logits; results are fp32 whatever the logits'
    dtype (for bf16 logits this equals the reference run on `logits.float()`).
    `materialize()` builds the tensor the reference would have built.
    """

    def __init__(self, logits: Tensor, temperature: float = 1.0):
        if logits.ndim != 3:
            raise ValueError(f"logits must be [batch, seq, vocab], got {tuple(logits.shape)}")
        self.logits = logits
        self.temperature = float(temperature)
        self._entropy: Tensor | None = None

    @property
    def shape(self):
        return self.logits.shape

    @property
    def device(self):
        return self.logits.device

    def __truediv__(self, temperature: float) -> "ShiftedLogits":
        return ShiftedLogits(self.logits, self.temperature * float(temperature))

    def materialize(self) -> Tensor:
        x = self.logits.float()
        out = torch.zeros_like(x)
        out[:, 1:, :] = x[:, :-1, :]
        return out / self.temperature


def shift_logits(logits: Tensor) -> ShiftedLogits:
    """Removes final token logits and adds a zero logit for the first token
    (src/prime_rl/trainer/rl/loss.py:51-58), without copying anything."""
    return ShiftedLogits(logits)


# --------------------------------------------------------------------------- kernel launches


def _rows(logits: Tensor) -> Tensor:
    if logits.stride(-1) != 1:
        logits = logits.contiguous()
    return logits


def _launch_fwd(
    logits: Tensor, index: Tensor, temperature: float, shift: bool, want_entropy: bool = True
) -> tuple[Tensor, Tensor | None, Tensor]:
    """logits [B, T, V] -> logprob, entropy, lse, each fp32 [B, T]."""
    dev = C.require_device(logits, index)
    lib = C.load()
    b, t, v = logits.shape
    if index.shape != (b, t):
        raise ValueError(f"index must be [batch, seq] = {(b, t)}, got {tuple(index.shape)}")
    if t > 0 and logits.stride(1) < v:
        logits = logits.contiguous()
    ids = index.to(torch.int64).contiguous()
    lp = torch.empty((b, t), dtype=torch.float32, device=dev)
    ent = torch.empty((b, t), dtype=torch.float32, device=dev) if want_entropy else None
    lse = torch.empty((b, t), dtype=torch.float32, device=dev)
    ws_bytes = lib.prl_logprob_workspace_bytes(t, v)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    code = C.dtype_code(logits)
    st = C.stream_ptr(dev)
    with C.on_device(dev), C.timed("prl_logprob_entropy_fwd", dev):
        for i in range(b):
            C.check(
                lib.prl_logprob_entropy_fwd(
                    C.ptr(logits[i]), code, t, v, logits.stride(1), C.ptr(ids[i]), temperature, int(shift),
                    C.ptr(lp[i]), C.ptr(ent[i]) if want_entropy else C.ptr(None), C.ptr(lse[i]),
                    C.ptr(ws), ws_bytes, st,
                ),
                "prl_logprob_entropy_fwd",
            )
    return lp, ent, lse


def _launch_bwd(
    logits: Tensor, index: Tensor, lse: Tensor, g: Tensor, gscale: Tensor | None, temperature: float, shift: bool
) -> Tensor:
    """dlogits [B, T, V] in the dtype of the logits."""
    dev = C.require_device(logits, index, lse, g, gscale)
    lib = C.load()
    b, t, v = logits.shape
    out = torch.empty((b, t, v), dtype=logits.dtype, device=dev)
    g = g.to(torch.float32).contiguous()
    code = C.dtype_code(logits)
    st = C.stream_ptr(dev)
    with C.on_device(dev), C.timed("prl_logprob_bwd", dev):
        for i in range(b):
            C.check(
                lib.prl_logprob_bwd(
                    C.ptr(logits[i]), code, t, v, logits.stride(1), C.ptr(index[i]), C.ptr(lse[i]), C.ptr(g[i]),
                    C.ptr(gscale), temperature, int(shift), C.ptr(out[i]), code, v, st,
                ),
                "prl_logprob_bwd",
            )
    return out


class _SelectiveLogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, index: Tensor, temperature: float, shift: bool):
        logits = _rows(logits)
        lp, ent, lse = _launch_fwd(logits, index, temperature, shift)
        ids = index.to(torch.int64).contiguous()
        ctx.save_for_backward(logits, ids, lse)
        ctx.temperature, ctx.shift = temperature, shift
        ctx.mark_non_differentiable(ent)
        ctx.set_materialize_grads(False)  # or autograd fills a zero gradient for the entropy on every backward
        return lp, ent

    @staticmethod
    def backward(ctx, grad_lp: Tensor | None, _grad_ent):
        if grad_lp is None:
            return None, None, None, None
        logits, ids, lse = ctx.saved_tensors
        return _launch_bwd(logits, ids, lse, grad_lp, None, ctx.temperature, ctx.shift), None, None, None


def selective_log_softmax(logits: Tensor | ShiftedLogits, index: Tensor) -> Tensor:
    """log softmax(logits)[index] per position, differentiable with respect to
    the logits (src/prime_rl/trainer/rl/loss.py:13-39).

    A `ShiftedLogits` argument is scored with its shift and temperature applied
    on the fly, in fp32.  A plain tensor is taken as already shifted and scaled;
    a plain bf16 tensor gives bf16 results, as the reference's low-precision
    branch does."""
    if isinstance(logits, ShiftedLogits):
        lp, ent = _SelectiveLogSoftmax.apply(logits.logits, index, logits.temperature, True)
        logits._entropy = ent
        return lp
    lp, _ = _SelectiveLogSoftmax.apply(logits, index, 1.0, False)
    return lp if logits.dtype == torch.float32 else lp.to(logits.dtype)


def compute_entropy(shifted_logits: Tensor | ShiftedLogits) -> Tensor:
    """H = logsumexp(x) - sum softmax(x) x (src/prime_rl/trainer/rl/loss.py:43-47).

    For a `ShiftedLogits` that `selective_log_softmax` has already seen, the
    entropy was produced by that same pass and is returned as is."""
    if isinstance(shifted_logits, ShiftedLogits):
        if shifted_logits._entropy is not None:
            return shifted_logits._entropy
        x, temp, shift = shifted_logits.logits, shifted_logits.temperature, True
    else:
        x, temp, shift = shifted_logits, 1.0, False
    x = _rows(x.detach())
    dummy = torch.zeros(x.shape[:2], dtype=torch.int64, device=x.device)
    _, ent, _ = _launch_fwd(x, dummy, temp, shift)
    return ent


# --------------------------------------------------------------------------- ratio-clip loss


def _ratio_code(loss_config: Any) -> int:
    if loss_config.ratio_type == "token":
        return C.RATIO_TOKEN
    if loss_config.ratio_type == "sequence":
        return C.RATIO_SEQUENCE
    raise ValueError(f"unknown ratio_type {loss_config.ratio_type!r}")


def _launch_loss(
    lp: Tensor, old: Tensor, adv: Tensor, mask: Tensor, seg_start: Tensor, n_seg: Tensor, max_seg: int,
    loss_config: Any, loss_scale: float,
) -> tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Flat fp32 [T] inputs -> loss [1], ratio, clipped, is_clipped, g."""
    dev = C.require_device(lp, old, adv, mask, seg_start, n_seg)
    lib = C.load()
    t = lp.numel()
    per_seq = _ratio_code(loss_config) == C.RATIO_SEQUENCE
    n_aux = max_seg if per_seq else t
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ratio = torch.empty(n_aux, dtype=torch.float32, device=dev)
    clipped = torch.empty(n_aux, dtype=torch.float32, device=dev)
    is_clipped = torch.empty(n_aux, dtype=torch.float32, device=dev)
    g = torch.empty(t, dtype=torch.float32, device=dev)
    seg_loss = torch.empty(max(max_seg, 1), dtype=torch.float32, device=dev)
    mask_u8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()
    with C.on_device(dev), C.timed("prl_grpo_loss", dev):
        C.check(
            lib.prl_grpo_loss(
                C.ptr(lp), C.ptr(old), C.ptr(adv), C.ptr(mask_u8), C.ptr(seg_start), C.ptr(n_seg), max_seg, t,
                _ratio_code(loss_config), int(bool(loss_config.ratio_length_norm)), float(loss_config.clip_ratio),
                float(loss_scale), C.ptr(loss), C.ptr(ratio), C.ptr(clipped), C.ptr(is_clipped), C.ptr(g),
                C.ptr(seg_loss), C.stream_ptr(dev),
            ),
            "prl_grpo_loss",
        )
    return loss, ratio, clipped, is_clipped, g


def _flat32(x: Tensor) -> Tensor:
    return x.reshape(-1).to(torch.float32).contiguous()


class _GrpoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp, old, adv, mask, seg_start, n_seg, max_seg, loss_config, loss_scale):
        loss, ratio, clipped, is_clipped, g = _launch_loss(
            lp.detach().contiguous(), old, adv, mask, seg_start, n_seg, max_seg, loss_config, loss_scale
        )
        ctx.save_for_backward(g)
        ctx.mark_non_differentiable(ratio, clipped, is_clipped)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the three logged tensors
        return loss.reshape(()), ratio, clipped, is_clipped

    @staticmethod
    def backward(ctx, grad_loss, *_):
        if grad_loss is None:
            return (None,) * 9
        (g,) = ctx.saved_tensors
        return (g * grad_loss,) + (None,) * 8


def compute_loss(
    logprobs: Sequence[Tensor],
    old_logprobs: Sequence[Tensor],
    advantages: Sequence[Tensor],
    loss_mask: Sequence[Tensor],
    loss_config: Any,
    loss_scale: float,
) -> tuple[Tensor, dict[str, Tensor]]:
    """Loss for packed sequences given as one tensor per sample
    (src/prime_rl/trainer/rl/loss.py:61-124).  All samples run in one launch;
    the returned scalar is differentiable with respect to `logprobs`."""
    logprobs, old_logprobs = list(logprobs), list(old_logprobs)
    advantages, loss_mask = list(advantages), list(loss_mask)
    lengths = [int(x.numel()) for x in logprobs]
    if not (len(old_logprobs) == len(advantages) == len(loss_mask) == len(lengths)):
        raise ValueError("compute_loss: the four sample lists must have the same length")
    n = len(lengths)
    if n == 0:
        raise ValueError("compute_loss: no samples given")
    dev = C.require_device(*logprobs)
    starts = [0]
    for k in lengths:
        starts.append(starts[-1] + k)
    seg_start = torch.tensor(starts, dtype=torch.int64).to(dev, non_blocking=True)
    n_seg = torch.tensor([n], dtype=torch.int32).to(dev, non_blocking=True)
    lp = torch.cat([x.reshape(-1) for x in logprobs]).float()
    old = _flat32(torch.cat([x.reshape(-1) for x in old_logprobs]))
    adv = _flat32(torch.cat([x.reshape(-1) for x in advantages]))
    mask = torch.cat([x.reshape(-1) for x in loss_mask])
    loss, ratio, clipped, is_clipped = _GrpoLoss.apply(
        lp, old, adv, mask, seg_start, n_seg, n, loss_config, loss_scale
    )
    return loss, {
        "importance_ratio": ratio,
        "clipped_importance_ratio": clipped,
        "is_clipped": is_clipped,
    }


# --------------------------------------------------------------------------- the whole epilogue as one call


def segments_from_position_ids(position_ids: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """seg_id int32 [T], seg_start int64 [max(T, 1) + 1], n_seg int32 [1], all on the
    device and without a host round trip (rule of src/prime_rl/trainer/utils.py:36-69; an empty row is one
    sample of length 0 there, so n_seg is never 0)."""
    dev = C.require_device(position_ids)
    lib = C.load()
    pos = position_ids.reshape(-1).to(torch.int64).contiguous()
    t = pos.numel()
    seg_id = torch.empty(t, dtype=torch.int32, device=dev)
    seg_start = torch.empty(max(t, 1) + 1, dtype=torch.int64, device=dev)
    n_seg = torch.empty(1, dtype=torch.int32, device=dev)
    with C.on_device(dev), C.timed("prl_segments_from_position_ids", dev):
        C.check(
            lib.prl_segments_from_position_ids(
                C.ptr(pos), t, C.ptr(seg_id), C.ptr(seg_start), C.ptr(n_seg), C.stream_ptr(dev)
            ),
            "prl_segments_from_position_ids",
        )
    return seg_id, seg_start, n_seg


def _u8(mask: Tensor) -> Tensor:
    m = mask.reshape(-1)
    return m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8).contiguous()


_UNIT_GRADIENTS: dict[torch.device, Tensor] = {}


def unit_gradient(device: torch.device | str) -> Tensor:
    """A constant 1.0 on `device` to pass to `loss.backward(...)`.

    `loss.backward()` without an argument makes autograd create a fresh tensor of ones whose VALUE the
    backward of `policy_loss` cannot know without a host synchronisation, so it launches a (tiny) kernel that
    looks at it on the device.  When the gradient that arrives IS this constant (same storage), that launch
    is skipped.  Do not modify the returned tensor."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    t = _UNIT_GRADIENTS.get(dev)
    if t is None:
        t = _UNIT_GRADIENTS[dev] = torch.ones((), dtype=torch.float32, device=dev)
    return t


def _seq_rows_limits() -> tuple[int, int] | None:
    """(longest sample whose workgroups wait for each other, sweeps a workgroup waits) of the one-pass
    sequence-mode kernel, or None when it is switched off (PRL_SEQ_ROWS=0).  PRL_SEQ_ROWS_MAX_LEN (224: a launch
    has one workgroup per compute unit, 256 of them, resident at a time) and PRL_SEQ_ROWS_SPINS (512 sweeps of one
    to two microseconds: with nothing else on the GPU every wait ends within 64-256 sweeps, and a workgroup whose
    sample cannot become resident, e.g. beside another process's kernels, gives its compute unit up after under a
    millisecond) override the defaults; rows the kernel leaves out go through the second-pass kernel."""
    import os

    if os.environ.get("PRL_SEQ_ROWS", "1") == "0":
        return None
    return int(os.environ.get("PRL_SEQ_ROWS_MAX_LEN", "224")), int(os.environ.get("PRL_SEQ_ROWS_SPINS", "512"))


def _rows_kernel_enabled() -> bool:
    """The register-resident single-pass kernel is used whenever it applies; PRL_FUSED_ROWS=0 switches it off
    (for A/B measurements against the two-kernel path)."""
    import os

    return os.environ.get("PRL_FUSED_ROWS", "1") != "0"


class _PolicyLoss(torch.autograd.Function):
    """Token mode with bf16 logits that need a gradient: ONE pass over the
    logits (prl_policy_rows_fwd_bwd) yields log-probabilities, entropy and the
    gradient row by row; the backward only hands that gradient out (rescaled on
    the device if the incoming loss gradient is not 1).
    Otherwise: forward prl_policy_forward, backward prl_logprob_bwd."""

    @staticmethod
    def forward(ctx, logits, input_ids, position_ids, advantages, old_logprobs, loss_mask, temperature,
                loss_config, loss_scale, skip_masked, positions=None, max_sample_len=None):
        dev = C.require_device(logits, input_ids, position_ids, advantages, old_logprobs, loss_mask)
        lib = C.load()
        logits = _rows(logits)
        _, t, v = logits.shape
        if t > 0 and logits.stride(1) < v:
            logits = logits.contiguous()
        n_rows = t  # rows of logits; with gathered rows T is the length of the packed row they were taken from
        if positions is not None:
            C.require_device(positions)
            positions = positions.reshape(-1).to(torch.int64).contiguous()
            if positions.numel() != n_rows:
                raise ValueError("policy_loss: `positions` must name one position per row of the logits")
            t = input_ids.numel()
        ids = input_ids.reshape(-1).to(torch.int64).contiguous()
        pos = position_ids.reshape(-1).to(torch.int64).contiguous()
        adv, old, mask = _flat32(advantages), _flat32(old_logprobs), _u8(loss_mask)
        if not (ids.numel() == pos.numel() == adv.numel() == old.numel() == mask.numel() == t):
            raise ValueError("policy_loss: every per-token tensor must hold T entries")
        ratio_code = _ratio_code(loss_config)
        per_seq = ratio_code == C.RATIO_SEQUENCE
        length_norm, clip = int(bool(loss_config.ratio_length_norm)), float(loss_config.clip_ratio)
        # fp32 arena: logprob, entropy, lse, g, ratio, clipped, is_clipped, seg_loss (T each) and the loss;
        # int64 arena: seg_start (T + 1) and n_seg (int32 in the last slot)
        f = torch.empty(8 * t + 1, dtype=torch.float32, device=dev)
        iarena = torch.empty(t + 2, dtype=torch.int64, device=dev)
        fp, ip = f.data_ptr(), iarena.data_ptr()
        at = lambda k: C.c_void_p(fp + 4 * t * k)
        code, st = C.dtype_code(logits), C.stream_ptr(dev)
        rows_kernel = (
            not per_seq and t > 0 and _rows_kernel_enabled() and logits.data_ptr() % 16 == 0
            and bool(lib.prl_policy_rows_supported(code, max(n_rows, 1), v, logits.stride(1) if n_rows else v, code, v))
        )
        single_pass = rows_kernel and ctx.needs_input_grad[0]
        # gathered rows nobody wants a gradient of: the same kernel, scoring only
        gathered_forward = rows_kernel and positions is not None and not ctx.needs_input_grad[0]
        if positions is not None and not (single_pass or gathered_forward):
            raise ValueError("policy_loss: gathered rows (`positions`) need ratio_type 'token', bf16 logits, a "
                             "vocabulary that is a multiple of 8 and at most 155648, and T > 0")
        seq_limits = _seq_rows_limits() if per_seq else None
        if seq_limits is not None and max_sample_len is not None and int(max_sample_len) > seq_limits[0]:
            # a sample too long for its workgroups to wait for each other: its rows would all be left to the second
            # pass, after the one-pass kernel has read them for nothing; two passes from the start are faster then
            # (and with a fifth of the rows left over the two passes already win)
            seq_limits = None
        seq_pass = (
            per_seq and seq_limits is not None and t > 0 and ctx.needs_input_grad[0] and _rows_kernel_enabled()
            and logits.data_ptr() % 16 == 0
            and bool(lib.prl_policy_rows_supported(code, t, v, logits.stride(1), code, v))
        )
        if seq_pass:
            # one ratio per sample, still one pass over the logits: the workgroups of a sample exchange their
            # log-ratios inside the launch; rows the launch leaves out (`pending`: samples too long to be resident
            # together, waits that ran out) are written by a second pass over those rows only
            dlogits = torch.empty((1, t, v), dtype=logits.dtype, device=dev)
            pending = torch.empty(t, dtype=torch.uint8, device=dev)
            ws_bytes = int(lib.prl_policy_rows_seq_workspace_bytes(t))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            with C.on_device(dev):
                with C.timed("prl_policy_rows_seq_fwd_bwd", dev):
                    C.check(
                        lib.prl_policy_rows_seq_fwd_bwd(
                            C.ptr(logits), code, t, v, logits.stride(1), C.ptr(ids), C.ptr(pos), C.ptr(adv),
                            C.ptr(old), C.ptr(mask), int(bool(skip_masked)), temperature, length_norm, clip,
                            float(loss_scale), at(0), at(1), at(2), C.ptr(dlogits), code, v, C.ptr(pending),
                            seq_limits[0], seq_limits[1], C.ptr(ws), ws_bytes, st,
                        ),
                        "prl_policy_rows_seq_fwd_bwd",
                    )
                with C.timed("prl_policy_tail", dev):
                    C.check(
                        lib.prl_policy_tail(
                            C.ptr(pos), at(0), C.ptr(old), C.ptr(adv), C.ptr(mask), t, ratio_code, length_norm, clip,
                            float(loss_scale), at(8), at(4), at(5), at(6), at(3), at(7), None, C.c_void_p(ip),
                            C.c_void_p(ip + 8 * (t + 1)), st,
                        ),
                        "prl_policy_tail",
                    )
                with C.timed("prl_logprob_bwd_selected", dev):
                    C.check(
                        lib.prl_logprob_bwd_selected(
                            C.ptr(logits), code, t, v, logits.stride(1), C.ptr(ids), at(2), at(3), None, temperature,
                            1, C.ptr(dlogits), code, v, C.ptr(pending), st,
                        ),
                        "prl_logprob_bwd_selected",
                    )
            ctx.dlogits = dlogits
            ctx.seq_pending = pending  # which rows the second pass wrote: `policy_loss` hands it out in `aux`
        elif positions is not None:
            # the LM head ran only where the next token is scored: row r of the logits belongs to position
            # positions[r]; every per-token result still comes back in the packed row's layout
            dlogits = torch.empty((1, n_rows, v), dtype=logits.dtype, device=dev) if single_pass else None
            with C.on_device(dev):
                with C.timed("prl_policy_rows_gathered_fwd_bwd", dev):
                    C.check(
                        lib.prl_policy_rows_gathered_fwd_bwd(
                            C.ptr(logits) if n_rows else None, code, n_rows, v, logits.stride(1) if n_rows else v,
                            C.ptr(positions) if n_rows else None, t, C.ptr(ids), C.ptr(adv), C.ptr(old), C.ptr(mask),
                            temperature, clip, float(loss_scale), at(0), at(1), at(2),
                            C.ptr(dlogits) if n_rows and single_pass else None, code, v, st,
                        ),
                        "prl_policy_rows_gathered_fwd_bwd",
                    )
                with C.timed("prl_policy_tail", dev):
                    C.check(
                        lib.prl_policy_tail(
                            C.ptr(pos), at(0), C.ptr(old), C.ptr(adv), C.ptr(mask), t, ratio_code, length_norm, clip,
                            float(loss_scale), at(8), at(4), at(5), at(6), at(3), at(7), None, C.c_void_p(ip),
                            C.c_void_p(ip + 8 * (t + 1)), st,
                        ),
                        "prl_policy_tail",
                    )
            ctx.dlogits = dlogits
        elif single_pass:
            dlogits = torch.empty((1, t, v), dtype=logits.dtype, device=dev)
            with C.on_device(dev):
                with C.timed("prl_policy_rows_fwd_bwd", dev):
                    C.check(
                        lib.prl_policy_rows_fwd_bwd(
                            C.ptr(logits), code, t, v, logits.stride(1), C.ptr(ids), C.ptr(adv), C.ptr(old),
                            C.ptr(mask), int(bool(skip_masked)), temperature, clip, float(loss_scale), at(0), at(1),
                            at(2), C.ptr(dlogits), code, v, st,
                        ),
                        "prl_policy_rows_fwd_bwd",
                    )
                with C.timed("prl_policy_tail", dev):
                    C.check(
                        lib.prl_policy_tail(
                            C.ptr(pos), at(0), C.ptr(old), C.ptr(adv), C.ptr(mask), t, ratio_code, length_norm, clip,
                            float(loss_scale), at(8), at(4), at(5), at(6), at(3), at(7), None, C.c_void_p(ip),
                            C.c_void_p(ip + 8 * (t + 1)), st,
                        ),
                        "prl_policy_tail",
                    )
            ctx.dlogits = dlogits
        else:
            ws_bytes = lib.prl_logprob_workspace_bytes(t, v)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
            with C.on_device(dev), C.timed("prl_policy_forward", dev):
                C.check(
                    lib.prl_policy_forward(
                        C.ptr(logits), code, t, v, logits.stride(1), C.ptr(ids), C.ptr(pos), C.ptr(adv),
                        C.ptr(old), C.ptr(mask), int(bool(skip_masked)), temperature, ratio_code, length_norm, clip,
                        float(loss_scale), at(0), at(1), at(2), at(8), at(4), at(5), at(6), at(3), at(7), None,
                        C.c_void_p(ip), C.c_void_p(ip + 8 * (t + 1)), C.ptr(ws), ws_bytes, st,
                    ),
                    "prl_policy_forward",
                )
        parts = f[: 8 * t].view(8, t)
        lp, ent, lse, g = parts[0].view(1, t), parts[1].view(1, t), parts[2].view(1, t), parts[3].view(1, t)
        n_aux = (t // 2 + 1 if t > 0 else 0) if per_seq else t
        ratio, clipped, is_clipped = parts[4][:n_aux], parts[5][:n_aux], parts[6][:n_aux]
        seg_start = iarena[: t + 1]
        n_seg = iarena[t + 1 :].view(torch.int32)[:1]
        # either way the gradient exists already and backward hands it out (or, for rows that were only scored,
        # nothing asks for one)
        single_pass = single_pass or seq_pass or gathered_forward
        ctx.single_pass = single_pass
        if not single_pass:
            ctx.save_for_backward(logits, ids.view(1, t), lse, g)
        ctx.temperature = temperature
        ctx.mark_non_differentiable(lp, ent, ratio, clipped, is_clipped, seg_start, n_seg)
        # without this autograd launches seven fill kernels per backward (a zero "gradient" for each of the
        # outputs above): 31 us per micro-batch next to a 220 us row kernel
        ctx.set_materialize_grads(False)
        return f[8 * t].reshape(()), lp, ent, ratio, clipped, is_clipped, seg_start, n_seg

    @staticmethod
    def backward(ctx, grad_loss, *_):
        if grad_loss is None:
            return (None,) * 12
        gscale = grad_loss.to(torch.float32).reshape(1).contiguous()
        if ctx.single_pass:
            d = ctx.dlogits
            if d is None:
                raise RuntimeError("policy_loss: the gradient of this loss has already been handed out; "
                                   "a second backward through the same forward is not supported")
            ctx.dlogits = None
            dev = d.device
            unit = _UNIT_GRADIENTS.get(dev)
            if unit is not None and grad_loss.data_ptr() == unit.data_ptr():
                return (d,) + (None,) * 11  # the caller passed unit_gradient(): known to be 1
            _, t, v = d.shape
            with C.on_device(dev), C.timed("prl_scale_rows_if_not_one", dev):
                C.check(
                    C.load().prl_scale_rows_if_not_one(C.ptr(d), C.dtype_code(d), t, v, v, C.ptr(gscale),
                                                       C.stream_ptr(dev)),
                    "prl_scale_rows_if_not_one",
                )
            return (d,) + (None,) * 11
        logits, ids, lse, g = ctx.saved_tensors
        d = _launch_bwd(logits, ids, lse, g, gscale, ctx.temperature, True)
        return (d,) + (None,) * 11


def policy_loss(
    logits: Tensor,
    input_ids: Tensor,
    position_ids: Tensor,
    advantages: Tensor,
    old_logprobs: Tensor,
    loss_mask: Tensor,
    temperature: float,
    loss_config: Any,
    loss_scale: float,
    skip_masked: bool = False,
    positions: Tensor | None = None,
    max_sample_len: int | None = None,
) -> tuple[Tensor, dict[str, Tensor]]:
    """One micro-batch of the trainer's hot loop from the model's logits to a
    differentiable loss (src/prime_rl/trainer/rl/train.py:285-303) in one call
    and without any host synchronisation.

    logits [1, T, V] (bf16 or fp32, straight from the model); the other tensors
    [1, T] as the `MicroBatch` holds them.  Returns the loss and a dict with
    `logprobs`, `entropy` (fp32 [1, T]), the three ratio tensors of
    `compute_loss`, and the device-side sample table (`seg_start`, `n_seg`).
    In sequence mode the ratio tensors have room for T // 2 + 1 samples, of
    which the first `n_seg` are filled.  `skip_masked=True` does not score
    tokens whose mask is False (their `logprobs` and `entropy` are 0, their
    logits are never read); loss, gradient and masked statistics are the same.

    With `ratio_type = "token"`, bf16 logits that require a gradient and a
    vocabulary of at most 155648 the logits are read exactly once: the same
    pass that scores a row writes its gradient, and `.backward()` only hands
    it on.  The same holds for `ratio_type = "sequence"`: the workgroups of a sample exchange their log-ratios
    inside the launch (rows of samples longer than ~224 tokens, and rows whose wait ran out, take a second pass
    over those rows only).  In every other case `.backward()` runs one more pass over the logits.

    `max_sample_len` (sequence mode only, optional): the longest sample of the row if the caller knows it from the
    host side (the trainers do, from the row's position ids before they are copied to the device).  A row with a
    sample of more than ~224 tokens goes through the two passes at once instead of trying the one-pass kernel
    first (308 against 348 us at the benchmark shape); without the hint the one-pass kernel is tried.

    `positions` (int64 [R], on the device): the logits are GATHERED rows, [1, R, V]: the LM head was run only at
    these positions of the packed row (`scored_positions` picks the ones whose next token is scored), row r
    predicting token positions[r] + 1.  Everything per token still has T entries; tokens that no row predicts
    come back unscored, as `skip_masked=True` leaves masked tokens, and the gradient is [1, R, V].  Token mode
    with bf16 logits only; logits that need no gradient (a pass under `torch.no_grad()`) are only scored.  With prompts masked out this takes their share out of the LM head's three products
    and out of the pass over the logits; loss, gradient and masked statistics are unchanged."""
    if logits.ndim != 3 or logits.shape[0] != 1:
        raise ValueError(f"policy_loss takes one packed row, logits [1, T, V]; got {tuple(logits.shape)}")
    loss, lp, ent, ratio, clipped, is_clipped, seg_start, n_seg = _PolicyLoss.apply(
        logits, input_ids, position_ids, advantages, old_logprobs, loss_mask, float(temperature), loss_config,
        float(loss_scale), bool(skip_masked), positions, max_sample_len,
    )
    aux = {
        "logprobs": lp,
        "entropy": ent,
        "importance_ratio": ratio,
        "clipped_importance_ratio": clipped,
        "is_clipped": is_clipped,
        "seg_start": seg_start,
        "n_seg": n_seg,
    }
    # one ratio per sample, gradient formed in the one-pass kernel: which rows it left to the second pass
    # (uint8 [T] on the device; for tests and for looking at how often a row's wait runs out)
    pending = getattr(loss.grad_fn, "seq_pending", None)
    if pending is not None:
        aux["seq_pending"] = pending
    return loss, aux


def scored_positions(loss_mask: Tensor, multiple: int = 1) -> Tensor:
    """Positions of a packed row whose logits are needed: p such that token p + 1 is scored (loss_mask[p + 1]),
    ascending, as an int64 tensor on the mask's device (meant for a mask that is still on the host: no device
    synchronisation then).  Token 0 is predicted by no position (the reference scores it against zeros,
    src/prime_rl/trainer/rl/loss.py:51-58).

    `multiple` > 1 rounds the count up with positions whose token is NOT scored (they cost a zero gradient row
    each and change nothing), so that the LM head's products see few distinct row counts from step to step."""
    m = loss_mask.reshape(-1).to(torch.bool)
    keep = m[1:].clone()
    n = int(keep.sum())
    if multiple > 1 and n % multiple and n > 0:
        spare = (~keep).nonzero().reshape(-1)[: multiple - n % multiple]
        keep[spare] = True
    return keep.nonzero().reshape(-1)
