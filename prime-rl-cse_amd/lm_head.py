"""The LM head's three matrix products on this package's MFMA kernels (csrc/lm_head.hip).

The reference ends its forward with the model's final projection (src/prime_rl/trainer/model.py:251-254) and lets
autograd form the two products of its backward behind `loss.backward()` (src/prime_rl/trainer/rl/train.py:310),
all three through the BLAS library.  `lm_head(hidden, weight)` is `torch.nn.functional.linear(hidden, weight)` for
bf16 matrices with those three products run by `prl_lm_head_fwd`, `prl_lm_head_bwd_input` and
`prl_lm_head_bwd_weight`.  There is no fallback inside these functions: a shape the kernels do not take raises, and
callers that want the library product for such shapes ask `supported()` first.
"""

from __future__ import annotations

import functools
import math
import os
from ctypes import c_int64, c_size_t

import torch

from . import _capi
from ._capi import check, on_device, ptr, require_device, stream_ptr, timed


def supported(hidden: torch.Tensor, weight: torch.Tensor) -> bool:
    """Whether `lm_head(hidden, weight)` takes these matrices: bf16 on a HIP device, [.., H] and [V, H] with unit
    stride along H, and T, H, V multiples of 64."""
    if hidden.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or not hidden.is_cuda or not weight.is_cuda:
        return False
    if weight.dim() != 2 or hidden.dim() < 2 or hidden.shape[-1] != weight.shape[1]:
        return False
    t = hidden.numel() // hidden.shape[-1]
    v, h = weight.shape
    return bool(_capi.load().prl_lm_head_supported(_capi.PRL_BF16, t, h, v))


def _rows(x: torch.Tensor, what: str) -> torch.Tensor:
    """x as a 2-D matrix whose rows are 16-byte aligned with unit stride inside a row (a copy if it is not)."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1 or x2.stride(0) % 8 != 0 or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()
    if x2.dtype != torch.bfloat16:
        raise RuntimeError(f"{what} must be bfloat16, got {x2.dtype}")
    return x2


def _workspace(t: int, h: int, v: int, device: torch.device) -> tuple[torch.Tensor, int]:
    n = int(_capi.load().prl_lm_head_workspace_bytes(t, h, v))
    return torch.empty(max(n, 16), dtype=torch.uint8, device=device), n


def lm_head_fwd(hidden: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """logits[T, V] = hidden[T, H] . weight[V, H]^T (no autograd)."""
    dev = require_device(hidden, weight)
    x, w = _rows(hidden, "hidden"), _rows(weight, "weight")
    t, h = x.shape
    v = w.shape[0]
    out = torch.empty((t, v), dtype=torch.bfloat16, device=dev)
    with on_device(dev), timed("prl_lm_head_fwd", dev):
        check(_capi.load().prl_lm_head_fwd(ptr(x), c_int64(x.stride(0)), ptr(w), c_int64(w.stride(0)), ptr(out),
                                           c_int64(v), c_int64(t), c_int64(h), c_int64(v), stream_ptr(dev)),
              "prl_lm_head_fwd")
    return out


def lm_head_bwd_input(dlogits: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """dhidden[T, H] = dlogits[T, V] . weight[V, H]."""
    dev = require_device(dlogits, weight)
    g, w = _rows(dlogits, "dlogits"), _rows(weight, "weight")
    t, v = g.shape
    h = w.shape[1]
    out = torch.empty((t, h), dtype=torch.bfloat16, device=dev)
    ws, n = _workspace(t, h, v, dev)
    with on_device(dev), timed("prl_lm_head_bwd_input", dev):
        check(_capi.load().prl_lm_head_bwd_input(ptr(g), c_int64(g.stride(0)), ptr(w), c_int64(w.stride(0)), ptr(out),
                                                 c_int64(h), c_int64(t), c_int64(h), c_int64(v), ptr(ws), c_size_t(n),
                                                 stream_ptr(dev)),
              "prl_lm_head_bwd_input")
    return out


def lm_head_bwd_weight(dlogits: torch.Tensor, hidden: torch.Tensor) -> torch.Tensor:
    """dweight[V, H] = dlogits[T, V]^T . hidden[T, H]."""
    dev = require_device(dlogits, hidden)
    g, x = _rows(dlogits, "dlogits"), _rows(hidden, "hidden")
    t, v = g.shape
    h = x.shape[1]
    out = torch.empty((v, h), dtype=torch.bfloat16, device=dev)
    ws, n = _workspace(t, h, v, dev)
    with on_device(dev), timed("prl_lm_head_bwd_weight", dev):
        check(_capi.load().prl_lm_head_bwd_weight(ptr(g), c_int64(g.stride(0)), ptr(x), c_int64(x.stride(0)), ptr(out),
                                                  c_int64(h), c_int64(t), c_int64(h), c_int64(v), ptr(ws), c_size_t(n),
                                                  stream_ptr(dev)),
              "prl_lm_head_bwd_weight")
    return out


def plan(t: int, h: int, v: int) -> tuple[bool, bool, bool]:
    """Which of the three products (forward, input gradient, weight gradient) of a [t, h] x [v, h] LM head run on
    this package's kernels; the others go to the BLAS library through torch.

    Default (`PRL_LM_HEAD=auto`): the two backward products, where the kernels are ahead of the library at every
    recipe shape measured (input gradient 1.07-2.06x, weight gradient 1.22-1.29x; `tools/lm_head_shapes.sh`), and
    the forward product except for hidden sizes between 2048 and 4096 (1.03-1.06x at 1024, level at 2048, 1.05x at
    5120; 1-2 % behind the library at 3584).  `PRL_LM_HEAD=hip` / `lib` take all three from one side;
    `PRL_LM_HEAD_FWD`, `PRL_LM_HEAD_DX`, `PRL_LM_HEAD_DW` = 0 / 1 set one product."""
    mode = os.environ.get("PRL_LM_HEAD", "auto")
    if mode not in ("auto", "hip", "lib"):
        raise RuntimeError(f"PRL_LM_HEAD must be auto, hip or lib, got {mode!r}")
    takes = bool(_capi.load().prl_lm_head_supported(_capi.PRL_BF16, t, h, v))
    ours = {"auto": (h <= 2048 or h >= 4096, True, True), "hip": (True, True, True), "lib": (False, False, False)}[mode]
    if not takes and mode == "auto":
        ours = (False, False, False)
    out = []
    for flag, name in zip(ours, ("PRL_LM_HEAD_FWD", "PRL_LM_HEAD_DX", "PRL_LM_HEAD_DW")):
        e = os.environ.get(name)
        out.append(flag if e is None else e != "0")
    if any(out) and not takes:
        raise RuntimeError(f"the LM-head kernels do not take T={t} H={h} V={v} (multiples of 64 needed)")
    return out[0], out[1], out[2]


@functools.lru_cache(maxsize=None)
def _gemm_table() -> dict[str, dict[int, tuple[float, float, float]]]:
    """`gemm_table.json` beside this file: for y = x W^T with W [out, in], library time / kernel time of the three
    products at 4096, 8192 and 16384 rows of x, measured on an MI355X by `tools/body_gemms_ab.py` for the linear
    layers of the decoder blocks of the recipes' models."""
    import json
    from pathlib import Path

    path = Path(__file__).with_name("gemm_table.json")
    if not path.exists():  # (a copy of the package without its measurements: every layer stays with the library)
        return {}
    raw = json.loads(path.read_text())["ratios"]
    return {key: {int(t): tuple(r) for t, r in by_t.items()} for key, by_t in raw.items()}


# A product moves to the kernels where they were measured at least this much ahead: (forward, input gradient, weight
# gradient).  The forward's bar is higher: there kernels and library are close for most layers, and the library's
# time for one and the same shape moves by up to 10 % from one measurement to the next.
LAYER_MARGINS = (1.10, 1.04, 1.04)


@functools.lru_cache(maxsize=4096)
def _layer_plan(mode: str, t: int, n_in: int, n_out: int) -> tuple[bool, bool, bool]:
    takes = bool(_capi.load().prl_lm_head_supported(_capi.PRL_BF16, t, n_in, n_out))
    if mode == "hip":
        return (takes, takes, takes)
    by_t = _gemm_table().get(f"{n_in}x{n_out}")
    if not takes or by_t is None:
        return (False, False, False)
    nearest = min(by_t, key=lambda m: abs(math.log2(t / m)))
    if abs(math.log2(t / nearest)) > 0.5:  # (more than a factor 1.41 from anything measured)
        return (False, False, False)
    r = by_t[nearest]
    return (r[0] >= LAYER_MARGINS[0], r[1] >= LAYER_MARGINS[1], r[2] >= LAYER_MARGINS[2])


def layer_plan(t: int, n_in: int, n_out: int) -> tuple[bool, bool, bool]:
    """Which of the three products (forward, input gradient, weight gradient) of a decoder block's linear layer
    y = x W^T, x [t, n_in], W [n_out, n_in] (bf16), run on this package's kernels: the same three kernels as the LM
    head's, which is one more such layer.  Default (`PRL_LAYER_GEMMS=auto`): the products that `gemm_table.json` holds
    as measured at least 4 % (the forward: 10 %) ahead of the BLAS library at the nearest measured row count; layers
    and row counts nobody measured stay with the library.  `lib`: none; `hip`: every product the kernels take."""
    mode = os.environ.get("PRL_LAYER_GEMMS", "auto")
    if mode not in ("auto", "hip", "lib"):
        raise RuntimeError(f"PRL_LAYER_GEMMS must be auto, hip or lib, got {mode!r}")
    if mode == "lib" or t <= 0:
        return (False, False, False)
    return _layer_plan(mode, int(t), int(n_in), int(n_out))


class _LMHeadMixed(torch.autograd.Function):
    """`linear(hidden, weight)` with each of the three products on the side `plan` names."""

    @staticmethod
    def forward(ctx, hidden: torch.Tensor, weight: torch.Tensor, fwd: bool, dx: bool, dw: bool) -> torch.Tensor:
        ctx.save_for_backward(hidden, weight)
        ctx.ours = (dx, dw)
        if fwd:
            return lm_head_fwd(hidden, weight).view(*hidden.shape[:-1], weight.shape[0])
        return torch.nn.functional.linear(hidden, weight)

    @staticmethod
    def backward(ctx, grad: torch.Tensor):
        hidden, weight = ctx.saved_tensors
        dx, dw = ctx.ours
        gh = gw = None
        g2 = grad.reshape(-1, grad.shape[-1])
        if ctx.needs_input_grad[0]:
            gh = (lm_head_bwd_input(g2, weight) if dx else g2 @ weight).view(hidden.shape)
        if ctx.needs_input_grad[1]:
            x2 = hidden.reshape(-1, hidden.shape[-1])
            gw = lm_head_bwd_weight(g2, x2) if dw else g2.t() @ x2
        return gh, gw, None, None, None


def linear(hidden: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """The LM head as the trainer runs it: `torch.nn.functional.linear(hidden, weight)` with the products `plan`
    assigns to this package's kernels taken from them.  Matrices the kernels do not take (other dtypes, sharded
    weights, sizes that are not multiples of 64) go to torch whole."""
    if hasattr(weight, "to_local") or not supported(hidden, weight):
        return torch.nn.functional.linear(hidden, weight)
    t = hidden.numel() // hidden.shape[-1]
    fwd, dx, dw = plan(t, weight.shape[1], weight.shape[0])
    if not (fwd or dx or dw):
        return torch.nn.functional.linear(hidden, weight)
    return _LMHeadMixed.apply(hidden, weight, fwd, dx, dw)


class _LMHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        ctx.save_for_backward(hidden, weight)
        out = lm_head_fwd(hidden, weight)
        return out.view(*hidden.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, grad: torch.Tensor):
        hidden, weight = ctx.saved_tensors
        gh = gw = None
        if ctx.needs_input_grad[0]:
            gh = lm_head_bwd_input(grad, weight).view(hidden.shape)
        if ctx.needs_input_grad[1]:
            gw = lm_head_bwd_weight(grad, hidden)
        return gh, gw


def lm_head(hidden: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """`torch.nn.functional.linear(hidden, weight)` (no bias) with all three products on the MFMA kernels."""
    return _LMHead.apply(hidden, weight)
