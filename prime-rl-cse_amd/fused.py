"""Fused HIP kernels for the transformer blocks, switched on per model.

The reference's switch is `model.impl = "liger_kernel"`, which swaps Triton
kernels for RMSNorm, RoPE and SwiGLU into the HF model
(src/prime_rl/trainer/model.py:74-75, src/prime_rl/trainer/config.py:179-184).
`fuse_model(model)` is that switch here: it rebinds the `forward` of the
model's normalisation modules to `rms_norm` (one HIP launch forward and two
backward instead of ~22 small torch kernels) and the model family's
`apply_rotary_pos_emb` to one launch for query and key together (instead of
~14 forward and ~20 backward), and the gated MLPs' `silu(gate) * up` to one
launch each way (instead of 2 and 5).  Modules whose shape or dtype the
kernels do not cover keep their torch forward (on the same device; nothing
moves to the host).
"""

from __future__ import annotations

import types
from typing import Any

import torch
from torch import Tensor, nn

from . import _capi as C
from . import lm_head


def _row_layout(x: Tensor):
    """How the kernels walk the rows of x (last dimension contiguous): (rows, row stride, group, group stride),
    group = 0 for evenly pitched rows, or None when x is neither that nor groups of evenly pitched rows."""
    h = x.shape[-1]
    rows = x.numel() // h
    try:
        x2 = x.view(rows, h)
        return rows, (x2.stride(0) if rows > 1 else h), 0, 0
    except RuntimeError:
        pass
    if x.ndim >= 3:
        try:
            x3 = x.view(-1, x.shape[-2], h)  # e.g. [token][head][d] inside a wider token row
        except RuntimeError:
            return None
        es = x.element_size()
        if (x3.stride(0) * es) % 16 == 0 and (x3.stride(1) * es) % 16 == 0:
            return rows, x3.stride(1), x3.shape[1], x3.stride(0)
    return None


class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, eps: float):
        dev = C.require_device(x, weight)
        lib = C.load()
        h = x.shape[-1]
        layout = _row_layout(x) if x.stride(-1) == 1 else None
        if layout is None:
            x = x.contiguous()
            layout = _row_layout(x)
        rows, stride, group, group_stride = layout
        w = weight.contiguous()
        y = torch.empty((rows, h), dtype=x.dtype, device=dev)
        rstd = torch.empty(rows, dtype=torch.float32, device=dev)
        with C.on_device(dev), C.timed("prl_rmsnorm_fwd", dev):
            C.check(
                lib.prl_rmsnorm_fwd(C.ptr(x), stride, C.ptr(w), C.ptr(y), h, C.ptr(rstd), C.dtype_code(x), rows, h, eps,
                                    group, group_stride, C.stream_ptr(dev)),
                "prl_rmsnorm_fwd",
            )
        ctx.save_for_backward(x, w, rstd)
        ctx.layout = layout
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, w, rstd = ctx.saved_tensors
        dev = x.device
        lib = C.load()
        rows, stride, group, group_stride = ctx.layout
        h = x.shape[-1]
        g2 = gy.reshape(rows, h)
        if g2.stride(-1) != 1 or (rows > 1 and (g2.stride(0) * g2.element_size()) % 16) or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        gx = torch.empty((rows, h), dtype=x.dtype, device=dev)
        gw = torch.empty(h, dtype=w.dtype, device=dev)
        code = C.dtype_code(x)
        ws_bytes = lib.prl_rmsnorm_bwd_workspace_bytes(code, rows, h)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with C.on_device(dev), C.timed("prl_rmsnorm_bwd", dev):
            C.check(
                lib.prl_rmsnorm_bwd(C.ptr(x), stride, C.ptr(w), C.ptr(rstd), C.ptr(g2),
                                    g2.stride(0) if rows > 1 else h, C.ptr(gx), h, C.ptr(gw), code, rows, h,
                                    C.ptr(ws), ws_bytes, group, group_stride, C.stream_ptr(dev)),
                "prl_rmsnorm_bwd",
            )
        return gx.view(x.shape), gw, None


_NORM_SHAPES: dict[tuple[torch.dtype, int], bool] = {}


def rms_norm_supported(x: Tensor, weight: Tensor) -> bool:
    """Whether `rms_norm` takes these tensors: same dtype (bf16 or fp32), on a HIP device, rows 16-byte aligned."""
    if not (x.is_cuda and weight.is_cuda) or x.dtype != weight.dtype or x.dtype not in (torch.bfloat16, torch.float32):
        return False
    h = x.shape[-1]
    if weight.shape != (h,) or x.numel() == 0:
        return False
    es = x.element_size()
    if (h * es) % 16 or x.data_ptr() % 16 or weight.data_ptr() % 16 or x.stride(-1) != 1:
        return False
    if x.ndim >= 2 and x.shape[:-1].numel() > 1 and any((s * es) % 16 for s in x.stride()[:-1]):
        return False
    key = (x.dtype, h)
    ok = _NORM_SHAPES.get(key)
    if ok is None:
        ok = _NORM_SHAPES[key] = bool(C.load().prl_rmsnorm_supported(C.dtype_code(x), h))
    return ok


def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """y = weight * cast(float(x) * rsqrt(mean(float(x)^2, -1) + eps)), differentiable in x and weight."""
    return _RMSNorm.apply(x, weight, float(eps))


class _AddRMSNorm(torch.autograd.Function):
    """(x + branch, norm(x + branch)) in one launch; the backward adds the gradient of the sum's other uses to the
    normalisation's in its one launch.  Both as the separate kernels would round them."""

    @staticmethod
    def forward(ctx, x: Tensor, branch: Tensor, weight: Tensor, eps: float):
        dev = C.require_device(x, branch, weight)
        lib = C.load()
        h = x.shape[-1]
        rows = x.numel() // h
        x2, b2 = x.reshape(rows, h), branch.reshape(rows, h)
        w = weight.contiguous()
        total = torch.empty((rows, h), dtype=x.dtype, device=dev)
        y = torch.empty((rows, h), dtype=x.dtype, device=dev)
        rstd = torch.empty(rows, dtype=torch.float32, device=dev)
        with C.on_device(dev), C.timed("prl_add_rmsnorm_fwd", dev):
            C.check(lib.prl_add_rmsnorm_fwd(C.ptr(x2), x2.stride(0) if rows > 1 else h, C.ptr(b2),
                                            b2.stride(0) if rows > 1 else h, C.ptr(total), h, C.ptr(w), C.ptr(y), h,
                                            C.ptr(rstd), C.dtype_code(x), rows, h, eps, C.stream_ptr(dev)),
                    "prl_add_rmsnorm_fwd")
        ctx.save_for_backward(total, w, rstd)
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)  # an output nobody used arrives as None, not as a tensor of zeros
        return total.view(x.shape), y.view(x.shape)

    @staticmethod
    def backward(ctx, gtotal: Tensor | None, gy: Tensor | None):
        total, w, rstd = ctx.saved_tensors
        dev = total.device
        lib = C.load()
        rows, h = total.shape
        if gy is None:  # the normalised output went nowhere: only the other uses of the sum have a gradient
            return gtotal, gtotal, None, None

        def rows16(g: Tensor) -> Tensor:
            g = g.reshape(rows, h)
            if g.stride(-1) != 1 or (rows > 1 and (g.stride(0) * g.element_size()) % 16) or g.data_ptr() % 16:
                g = g.contiguous()
            return g

        g2 = rows16(gy)
        gx = torch.empty((rows, h), dtype=total.dtype, device=dev)
        gw = torch.empty(h, dtype=w.dtype, device=dev)
        code = C.dtype_code(total)
        ws_bytes = lib.prl_rmsnorm_bwd_workspace_bytes(code, rows, h)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with C.on_device(dev), C.timed("prl_add_rmsnorm_bwd", dev):
            if gtotal is None:
                C.check(lib.prl_rmsnorm_bwd(C.ptr(total), h, C.ptr(w), C.ptr(rstd), C.ptr(g2),
                                            g2.stride(0) if rows > 1 else h, C.ptr(gx), h, C.ptr(gw), code, rows, h,
                                            C.ptr(ws), ws_bytes, 0, 0, C.stream_ptr(dev)), "prl_rmsnorm_bwd")
            else:
                ga = rows16(gtotal)
                C.check(lib.prl_add_rmsnorm_bwd(C.ptr(total), h, C.ptr(w), C.ptr(rstd), C.ptr(g2),
                                                g2.stride(0) if rows > 1 else h, C.ptr(ga),
                                                ga.stride(0) if rows > 1 else h, C.ptr(gx), h, C.ptr(gw), code, rows,
                                                h, C.ptr(ws), ws_bytes, C.stream_ptr(dev)), "prl_add_rmsnorm_bwd")
        gx = gx.view(ctx.shape)
        return gx, gx, gw, None


def add_rms_norm_supported(x: Tensor, branch: Tensor, weight: Tensor) -> bool:
    """Whether `add_rms_norm` takes these tensors: what `rms_norm` takes, for two inputs of one shape."""
    return x.shape == branch.shape and x.dtype == branch.dtype and x.ndim >= 2 \
        and rms_norm_supported(x, weight) and rms_norm_supported(branch, weight) \
        and _row_layout(x) is not None and _row_layout(x)[2] == 0 \
        and _row_layout(branch) is not None and _row_layout(branch)[2] == 0


def add_rms_norm(x: Tensor, branch: Tensor, weight: Tensor, eps: float) -> tuple[Tensor, Tensor]:
    """(s, y) with s = x + branch (in their dtype) and y = rms_norm(s, weight, eps), differentiable in all three."""
    return _AddRMSNorm.apply(x, branch, weight, float(eps))


# --------------------------------------------------------------------------- rotary position embedding


def _strides6(x: Tensor, out: Tensor):
    return (C.c_int64 * 6)(x.stride(0), x.stride(1), x.stride(2), out.stride(0), out.stride(1), out.stride(2))


def _launch_rope(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor, inverse: bool) -> tuple[Tensor, Tensor]:
    dev = q.device
    lib = C.load()
    b, hq, t, d = q.shape
    qo, ko = torch.empty_like(q), torch.empty_like(k)  # same strides as the inputs
    with C.on_device(dev), C.timed("prl_rope", dev):
        C.check(
            lib.prl_rope(C.ptr(q), C.ptr(qo), _strides6(q, qo), hq, C.ptr(k), C.ptr(ko), _strides6(k, ko), k.shape[1],
                         C.ptr(cos), C.ptr(sin), cos.stride(0), cos.stride(1), C.dtype_code(q), b, t, d, int(inverse),
                         C.stream_ptr(dev)),
            "prl_rope",
        )
    return qo, ko


def _rope_ready(x: Tensor) -> Tensor:
    """Last dimension contiguous, every other stride and the base address 16-byte aligned; else one copy."""
    es = x.element_size()
    if x.stride(-1) != 1 or x.data_ptr() % 16 or any((s * es) % 16 for s in x.stride()[:-1]):
        return x.contiguous()
    return x


class _Rope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q: Tensor, k: Tensor, cos: Tensor, sin: Tensor):
        q, k = _rope_ready(q), _rope_ready(k)
        ctx.save_for_backward(cos, sin)
        return _launch_rope(q, k, cos, sin, False)

    @staticmethod
    def backward(ctx, gq: Tensor, gk: Tensor):
        cos, sin = ctx.saved_tensors
        gq, gk = _launch_rope(_rope_ready(gq), _rope_ready(gk), cos, sin, True)
        return gq, gk, None, None


def rope_supported(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> bool:
    if q.ndim != 4 or k.ndim != 4 or cos.ndim != 3 or cos.shape != sin.shape or not q.is_cuda:
        return False
    if not (q.dtype == k.dtype == cos.dtype == sin.dtype) or q.dtype not in (torch.bfloat16, torch.float32):
        return False
    b, _, t, d = q.shape
    if k.shape[0] != b or k.shape[2] != t or k.shape[3] != d or cos.shape != (b, t, d) or q.numel() == 0:
        return False
    es = q.element_size()
    for c in (cos, sin):
        if c.stride(-1) != 1 or c.data_ptr() % 16 or (c.stride(0) * es) % 16 or (c.stride(1) * es) % 16:
            return False
    if cos.stride() != sin.stride() or cos.requires_grad or sin.requires_grad:
        return False
    return bool(C.load().prl_rope_supported(C.dtype_code(q), d))


def apply_rotary_pos_emb(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor, unsqueeze_dim: int = 1):
    """transformers' function of the same name ([batch, heads, seq, head_dim] layout, `unsqueeze_dim = 1`) on one
    HIP launch for query and key together; other layouts go to the original."""
    if unsqueeze_dim == 1 and rope_supported(q, k, cos, sin):
        return _Rope.apply(q, k, cos, sin)
    return _ORIGINAL_ROPE[0](q, k, cos, sin, unsqueeze_dim)


# --------------------------------------------------------------------------- SwiGLU


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate: Tensor, up: Tensor):
        dev = gate.device
        out = torch.empty_like(gate)
        with C.on_device(dev), C.timed("prl_swiglu_fwd", dev):
            C.check(C.load().prl_swiglu_fwd(C.ptr(gate), C.ptr(up), C.ptr(out), C.dtype_code(gate), gate.numel(),
                                            C.stream_ptr(dev)), "prl_swiglu_fwd")
        ctx.save_for_backward(gate, up)  # the activation's output is recomputed in the backward, not kept
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        gate, up = ctx.saved_tensors
        dev = gate.device
        if not gout.is_contiguous() or gout.data_ptr() % 16:
            gout = gout.contiguous()
        ggate, gup = torch.empty_like(gate), torch.empty_like(up)
        with C.on_device(dev), C.timed("prl_swiglu_bwd", dev):
            C.check(C.load().prl_swiglu_bwd(C.ptr(gate), C.ptr(up), C.ptr(gout), C.ptr(ggate), C.ptr(gup),
                                            C.dtype_code(gate), gate.numel(), C.stream_ptr(dev)), "prl_swiglu_bwd")
        return ggate, gup


def swiglu_supported(gate: Tensor, up: Tensor) -> bool:
    if not gate.is_cuda or gate.dtype != up.dtype or gate.dtype not in (torch.bfloat16, torch.float32):
        return False
    if gate.shape != up.shape or not gate.is_contiguous() or not up.is_contiguous() or gate.numel() == 0:
        return False
    return gate.numel() % (16 // gate.element_size()) == 0 and gate.data_ptr() % 16 == 0 and up.data_ptr() % 16 == 0


def swiglu(gate: Tensor, up: Tensor) -> Tensor:
    """silu(gate) * up, differentiable in both."""
    return _SwiGLU.apply(gate, up)


# --------------------------------------------------------------------------- one product for adjacent weights


def adjacent_rows(*weights: Tensor) -> Tensor | None:
    """One [sum of rows, K] view over 2-D weights that lie back to back in ONE storage (as the parameters of a
    block do in `replica.FlatState`'s flat buffer), or None.  A product over it replaces one product per
    weight: at 2048 tokens the blocks' GEMMs are too small to fill 256 compute units one by one (q + k + v
    separately 44 us, together 23 us; their weight gradients 61 -> 29 us)."""
    w0 = weights[0]
    if w0.ndim != 2 or not w0.is_cuda:
        return None
    base = w0.untyped_storage().data_ptr()
    ptr, rows = w0.data_ptr(), 0
    for w in weights:
        if w.ndim != 2 or w.dtype != w0.dtype or w.shape[1] != w0.shape[1] or not w.is_contiguous() \
                or w.untyped_storage().data_ptr() != base or w.data_ptr() != ptr or hasattr(w, "to_local"):
            return None
        ptr += w.numel() * w.element_size()
        rows += w.shape[0]
    return torch.as_strided(w0.detach(), (rows, w0.shape[1]), (w0.shape[1], 1))


def _on_kernels(rows: int, n_in: int, n_out: int, *operands: Tensor) -> tuple[bool, bool, bool]:
    """Which of (forward, input gradient, weight gradient) of y = x W^T, x [rows, n_in], W [n_out, n_in], run on the
    package's MFMA kernels (`lm_head.layer_plan`) rather than the BLAS library."""
    if any(z.dtype != torch.bfloat16 or not z.is_cuda for z in operands):
        return (False, False, False)
    return lm_head.layer_plan(rows, n_in, n_out)


def mm_nt(x2: Tensor, w: Tensor) -> Tensor:
    """x2 [rows, in] . w [out, in]^T."""
    if _on_kernels(x2.shape[0], w.shape[1], w.shape[0], x2, w)[0]:
        return lm_head.lm_head_fwd(x2, w)
    return x2 @ w.t()


def mm_nn(gy: Tensor, w: Tensor) -> Tensor:
    """gy [rows, out] . w [out, in]."""
    if _on_kernels(gy.shape[0], w.shape[1], w.shape[0], gy, w)[1]:
        return lm_head.lm_head_bwd_input(gy, w)
    return gy @ w


def weight_grad(gy: Tensor, x: Tensor) -> Tensor:
    """dW = gy^T x for gy [rows, out], x [rows, in] (both contiguous), in gy's dtype.

    On the package's MFMA kernels where `lm_head.layer_plan` says they are ahead; otherwise on the BLAS library:

    With a long row (16 K tokens) this product has a small output and a long sum: 1024 x 2048 outputs are 128
    tiles for 256 compute units, and the BLAS library's pick runs at 0.44-0.87 PFLOP/s where the forward products
    reach 1.2.  Cutting the sum over the tokens into S batched products and adding the S partial results fills
    the chip: 164 -> 89 us (o), 181 -> 111 us (down), 217 -> 154 us (q/k/v), 244 -> 203 us (gate/up),
    `tools/gemm_dw_split.py`.  Small outputs keep their partial results in fp32 (no extra rounding); large ones
    in bf16, where fp32 partials would cost the time back (relative error 2.4e-3 against 1.7e-3 for one
    product, both from rounding to bf16).  The partial results are added by `prl_sum_parts` (in order, fp32, one
    rounding): 8.9 / 7.5 us for fp32 / bf16 parts where torch's reduction (+ conversion) took 14.3 + 4.9 / 14.9."""
    rows, n_out = gy.shape
    n_in = x.shape[1]
    if _on_kernels(rows, n_in, n_out, gy, x)[2]:
        return lm_head.lm_head_bwd_weight(gy, x)
    size = n_out * n_in
    s = 4 if size <= 5_000_000 else 2
    if rows < 8192 or rows % s or size > 8_000_000 or gy.dtype != torch.bfloat16:
        return gy.t() @ x
    a, b = gy.view(s, rows // s, n_out).transpose(1, 2), x.view(s, rows // s, n_in)
    parts = torch.bmm(a, b, out_dtype=torch.float32) if size <= 3_200_000 else torch.bmm(a, b)
    if size % 8 or not parts.is_contiguous():
        return parts.sum(0).to(gy.dtype)
    dev = gy.device
    out = torch.empty((n_out, n_in), dtype=gy.dtype, device=dev)
    with C.on_device(dev), C.timed("prl_sum_parts", dev):  # the partial results added in order, rounded once
        C.check(C.load().prl_sum_parts(C.ptr(parts), C.dtype_code(parts), size, s, size, C.ptr(out),
                                       C.stream_ptr(dev)), "prl_sum_parts")
    return out


class _Linear(torch.autograd.Function):
    """x W^T (+ bias), as nn.Linear computes it, with each of the three products on the side `lm_head.layer_plan`
    names; the library's weight gradient goes through `weight_grad`.  With a bias the forward stays the library's
    (which adds the bias inside its product) and the bias's gradient is the column sum autograd would form."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor | None = None):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if bias is None and x.dim() >= 2 \
                and _on_kernels(x.numel() // x.shape[-1], weight.shape[1], weight.shape[0], x, weight)[0]:
            return lm_head.lm_head_fwd(_rows2d(x), weight).view(*x.shape[:-1], weight.shape[0])
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, weight = ctx.saved_tensors
        g2 = _rows2d(gy)
        dx = mm_nn(g2, weight).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = weight_grad(g2, _rows2d(x)) if ctx.needs_input_grad[1] else None
        db = g2.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


def _split_dw_linear_forward(self: nn.Module, x: Tensor) -> Tensor:
    w, b = self.weight, self.bias
    if hasattr(w, "to_local") or not x.is_cuda or x.dtype != w.dtype or (b is not None and b.dtype != w.dtype) \
            or not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)):
        return self._prl_torch_forward(x)
    return _Linear.apply(x, w, b)


def _rows2d(x: Tensor) -> Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    return x2 if x2.is_contiguous() else x2.contiguous()


class _LinearSplit(torch.autograd.Function):
    """y = x W^T for W = the adjacent weights stacked, returned split per weight (views of one matrix); the
    backward joins the pieces' gradients and runs ONE product each for dx and for the stacked dW."""

    @staticmethod
    def forward(ctx, x: Tensor, stacked: Tensor, *weights: Tensor):
        x2 = _rows2d(x)
        y = mm_nt(x2, stacked)
        ctx.save_for_backward(x2, stacked)
        ctx.sizes = [w.shape[0] for w in weights]
        ctx.lead = x.shape[:-1]
        return tuple(p.view(*ctx.lead, p.shape[-1]) for p in y.split(ctx.sizes, dim=1))

    @staticmethod
    def backward(ctx, *grads: Tensor):
        x2, stacked = ctx.saved_tensors
        rows = x2.shape[0]
        parts = [torch.zeros((rows, n), dtype=x2.dtype, device=x2.device) if g is None else g.reshape(rows, n)
                 for g, n in zip(grads, ctx.sizes)]
        gy = torch.cat(parts, dim=1)
        dx = mm_nn(gy, stacked).view(*ctx.lead, x2.shape[1]) if ctx.needs_input_grad[0] else None
        dw = weight_grad(gy, x2).split(ctx.sizes, dim=0)
        return (dx, None) + tuple(dw)


class _GateUpSwiGLU(torch.autograd.Function):
    """silu(x Wg^T) * (x Wu^T) with ONE product over the stacked [Wg; Wu] and the activation kernels working on
    column blocks of its output; the backward writes both gradients into one matrix and runs one product each
    for dx and for the stacked dW."""

    @staticmethod
    def forward(ctx, x: Tensor, stacked: Tensor, wg: Tensor, wu: Tensor):
        dev = x.device
        x2 = _rows2d(x)
        gu = mm_nt(x2, stacked)
        rows, n = x2.shape[0], wg.shape[0]
        out = torch.empty((rows, n), dtype=x.dtype, device=dev)
        es = x.element_size()
        with C.on_device(dev), C.timed("prl_swiglu_rows_fwd", dev):
            C.check(C.load().prl_swiglu_rows_fwd(gu.data_ptr(), gu.data_ptr() + n * es, 2 * n, C.ptr(out), n,
                                                 C.dtype_code(x), rows, n, C.stream_ptr(dev)), "prl_swiglu_rows_fwd")
        ctx.save_for_backward(x2, gu, stacked)
        ctx.lead, ctx.n = x.shape[:-1], n
        return out.view(*ctx.lead, n)

    @staticmethod
    def backward(ctx, gout: Tensor):
        x2, gu, stacked = ctx.saved_tensors
        dev = x2.device
        rows, n = x2.shape[0], ctx.n
        g2 = gout.reshape(rows, n)
        if g2.stride(1) != 1 or g2.stride(0) % 8 or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        ggu = torch.empty_like(gu)
        es = gu.element_size()
        with C.on_device(dev), C.timed("prl_swiglu_rows_bwd", dev):
            C.check(C.load().prl_swiglu_rows_bwd(gu.data_ptr(), gu.data_ptr() + n * es, 2 * n, C.ptr(g2), g2.stride(0),
                                                 ggu.data_ptr(), ggu.data_ptr() + n * es, 2 * n, C.dtype_code(gu), rows,
                                                 n, C.stream_ptr(dev)), "prl_swiglu_rows_bwd")
        dx = mm_nn(ggu, stacked).view(*ctx.lead, x2.shape[1]) if ctx.needs_input_grad[0] else None
        dw = weight_grad(ggu, x2)
        return dx, None, dw[:n], dw[n:]


def _stackable(x: Tensor, *linears: nn.Module) -> Tensor | None:
    if any(getattr(m, "bias", None) is not None for m in linears) or not x.is_cuda:
        return None
    if x.dtype not in (torch.bfloat16, torch.float32) or x.dtype != linears[0].weight.dtype:
        return None
    return adjacent_rows(*(m.weight for m in linears))


def _fused_mlp_forward(self: nn.Module, x: Tensor) -> Tensor:
    wg, wu = self.gate_proj.weight, self.up_proj.weight
    stacked = _stackable(x, self.gate_proj, self.up_proj) if wg.shape == wu.shape else None
    if stacked is not None and wg.shape[0] % (16 // x.element_size()) == 0:
        return self.down_proj(_GateUpSwiGLU.apply(x, stacked, wg, wu))
    gate, up = self.gate_proj(x), self.up_proj(x)
    if swiglu_supported(gate, up):
        return self.down_proj(swiglu(gate, up))
    return self.down_proj(self.act_fn(gate) * up)


_ORIGINAL_ROPE: list[Any] = []
_PATCHED_MODULES: dict[str, Any] = {}


class _Carry:
    """What one block leaves for whoever reads its output next (the next block, or the final norm): its MLP
    output, not yet added to the hidden states it returned, so that the addition runs inside that reader's
    normalisation kernel.  Bound to the identity of the returned tensor: a reader that is handed anything else
    while a branch is pending fails instead of computing without it."""

    def __init__(self) -> None:
        self.hidden: Tensor | None = None
        self.branch: Tensor | None = None

    def clear(self) -> None:
        self.hidden = self.branch = None

    def put(self, hidden: Tensor, branch: Tensor) -> None:
        self.hidden, self.branch = hidden, branch

    def take(self, hidden: Tensor) -> Tensor | None:
        if self.branch is None:
            return None
        mine, branch = self.hidden is hidden, self.branch
        self.clear()
        if not mine:
            raise RuntimeError("fused blocks: the output of a block was replaced before the next block or the final "
                               "norm read it; the residual carry needs them called back to back "
                               "(fused.enable_residual_carry(model, False) switches it off)")
        return branch


def _add_norm(norm: nn.Module, x: Tensor, branch: Tensor) -> tuple[Tensor, Tensor]:
    """(x + branch, norm(x + branch)): one launch where the kernel takes the tensors, else the two steps."""
    w = norm.weight
    if not hasattr(w, "to_local") and add_rms_norm_supported(x, branch, w):
        return add_rms_norm(x, branch, w, norm.variance_epsilon)
    total = x + branch
    return total, norm(total)


def _fused_layer_forward(self: nn.Module, hidden_states: Tensor, attention_mask: Tensor | None = None,
                         position_ids: Tensor | None = None, past_key_values: Any = None,
                         use_cache: bool | None = False, position_embeddings: Any = None, **kwargs: Any) -> Tensor:
    """The forward of a transformers decoder layer (residual, norm, attention, add, residual, norm, MLP, add) with
    each addition inside the normalisation that follows it: the one after the attention always, the one after
    the MLP when the layers hand it on (`enable_residual_carry`)."""
    carry: _Carry | None = self._prl_carry
    x = hidden_states
    pending = None
    if carry is not None:
        if self._prl_first:
            carry.clear()  # nothing of an earlier, abandoned forward
        pending = carry.take(x)
    if pending is not None:
        x, normed = _add_norm(self.input_layernorm, x, pending)
    else:
        normed = self.input_layernorm(x)
    branch, _ = self.self_attn(hidden_states=normed, attention_mask=attention_mask, position_ids=position_ids,
                               past_key_values=past_key_values, use_cache=use_cache,
                               position_embeddings=position_embeddings, **kwargs)
    x, normed = _add_norm(self.post_attention_layernorm, x, branch)
    branch = self.mlp(normed)
    if carry is None:
        return x + branch
    carry.put(x, branch)
    return x


_LAYER_SHAPE = ("residual = hidden_states", "hidden_states = self.input_layernorm(hidden_states)",
                "hidden_states = residual + hidden_states",
                "hidden_states = self.post_attention_layernorm(hidden_states)",
                "hidden_states = self.mlp(hidden_states)", "return hidden_states")


def _is_plain_decoder_layer(module: nn.Module) -> bool:
    """A decoder layer whose own forward is exactly the sequence `_fused_layer_forward` restates (Qwen3, Qwen2 and
    Llama layers of this transformers version are; anything that adds a step is left alone)."""
    if not type(module).__name__.endswith("DecoderLayer"):
        return False
    if not all(hasattr(module, a) for a in ("input_layernorm", "post_attention_layernorm", "self_attn", "mlp")):
        return False
    if not all(hasattr(n, "variance_epsilon") and hasattr(n, "weight")
               for n in (module.input_layernorm, module.post_attention_layernorm)):
        return False
    import inspect

    try:
        lines = [ln.strip() for ln in inspect.getsource(type(module).forward).splitlines()]
    except (OSError, TypeError):
        return False
    body = [ln for ln in lines if ln and not ln.startswith("#")]
    pos = 0
    for want in _LAYER_SHAPE:  # in this order, and the two additions both present
        try:
            pos = body.index(want, pos) + 1
        except ValueError:
            return False
    return body.count("hidden_states = residual + hidden_states") == 2 and body.count("residual = hidden_states") == 2


def enable_residual_carry(model: nn.Module, on: bool = True) -> int:
    """Let every fused decoder layer hand its MLP output to the next layer's input norm (and the last layer's
    to the final norm) instead of adding it itself.  Only for a model whose layers run back to back in one
    process-local forward: not under per-layer activation checkpointing (a re-run layer would not get the branch
    again) nor under per-layer FSDP (its pre-backward hook hangs on the layer's returned tensor, and the MLP's
    backward can run before it).  Returns the number of layers that carry (0: the model is not shaped for it)."""
    inner = getattr(model, "model", None)
    layers = list(getattr(inner, "layers", []) or [])
    final = getattr(inner, "norm", None)
    ready = bool(layers) and final is not None and hasattr(final, "_prl_torch_forward") \
        and all(hasattr(m, "_prl_carry") for m in layers)
    if not on or not ready:
        for m in [*layers, final]:
            if m is not None and hasattr(m, "_prl_carry"):
                m._prl_carry = None
        return 0
    carry = _Carry()
    for i, m in enumerate(layers):
        m._prl_carry, m._prl_first = carry, i == 0
    final._prl_carry = carry
    return len(layers)


def _fused_rmsnorm_forward(self: nn.Module, hidden_states: Tensor) -> Tensor:
    carry: _Carry | None = getattr(self, "_prl_carry", None)
    if carry is not None:  # the final norm behind layers that carry their last addition
        pending = carry.take(hidden_states)
        if pending is not None:
            return _add_norm(self, hidden_states, pending)[1]
    weight = self.weight
    if hasattr(weight, "to_local"):  # a DTensor outside an FSDP forward: leave it to torch
        return self._prl_torch_forward(hidden_states)
    if rms_norm_supported(hidden_states, weight):
        return rms_norm(hidden_states, weight, self.variance_epsilon)
    return self._prl_torch_forward(hidden_states)


def _as_one_matrix(gq: Tensor | None, gk: Tensor | None, gv: Tensor | None, t: int, hq: int, hk: int, d: int,
                   like: Tensor) -> Tensor | None:
    """[t, (hq + 2 hk) d] over the memory of gq, gk, gv ([1, heads, t, d] each) when they are the query, key and
    value blocks of exactly such a matrix, in that order, in `like`'s dtype and row width; else None."""
    if gq is None or gk is None or gv is None:
        return None
    width = (hq + 2 * hk) * d
    if like.shape != (t, width) or any(g.dtype != like.dtype or g.device != like.device for g in (gq, gk, gv)):
        return None
    es = like.element_size()
    at = gq.data_ptr()
    for g, heads in ((gq, hq), (gk, hk), (gv, hk)):
        if g.shape != (1, heads, t, d) or g.stride()[1:] != (d, width, 1) or g.data_ptr() != at \
                or g.untyped_storage().data_ptr() != gq.untyped_storage().data_ptr():
            return None
        at += heads * d * es
    if gq.data_ptr() % 16:
        return None
    return torch.as_strided(gq, (t, width), (width, 1), gq.storage_offset())


class _QKVNormRope(torch.autograd.Function):
    """Everything a Qwen3 attention block does between its input and its attention call, for one packed row:
    ONE product over the stacked q/k/v weights, one kernel for the query / key RMSNorm and the rotary embedding;
    backward: one kernel that writes the query and key gradients into the stacked gradient matrix, the value
    gradient copied beside them, ONE product each for dx and the stacked dW."""

    @staticmethod
    def forward(ctx, x, stacked, wq, wk, wv, nq, nk, cos, sin, eps: float):
        dev = x.device
        lib = C.load()
        x2 = _rows2d(x)
        t, d = x2.shape[0], 128
        hq, hk = wq.shape[0] // d, wk.shape[0] // d
        y = mm_nt(x2, stacked)
        q = torch.empty((t, hq, d), dtype=x.dtype, device=dev)
        k = torch.empty((t, hk, d), dtype=x.dtype, device=dev)
        rstd = torch.empty((t, hq + hk), dtype=torch.float32, device=dev)
        cos2, sin2 = cos.reshape(t, d), sin.reshape(t, d)
        with C.on_device(dev), C.timed("prl_qk_norm_rope_fwd", dev):
            C.check(lib.prl_qk_norm_rope_fwd(C.ptr(y), y.stride(0), C.ptr(nq), C.ptr(nk), C.ptr(cos2), C.ptr(sin2),
                                             cos2.stride(0), C.ptr(q), C.ptr(k), C.ptr(rstd), C.dtype_code(x), t, hq,
                                             hk, eps, C.stream_ptr(dev)), "prl_qk_norm_rope_fwd")
        ctx.save_for_backward(x2, stacked, y, rstd, nq, nk, cos2, sin2)
        ctx.lead, ctx.heads = x.shape[:-1], (hq, hk)
        ctx.sizes = [wq.shape[0], wk.shape[0], wv.shape[0]]
        v = y[:, wq.shape[0] + wk.shape[0]:]
        # [1, heads, T, d] views, the layout the attention functions take
        return (q.view(1, t, hq, d).transpose(1, 2), k.view(1, t, hk, d).transpose(1, 2),
                v.view(1, t, -1, d).transpose(1, 2))

    @staticmethod
    def backward(ctx, gq, gk, gv):
        x2, stacked, y, rstd, nq, nk, cos2, sin2 = ctx.saved_tensors
        dev = x2.device
        lib = C.load()
        t, d = x2.shape[0], 128
        hq, hk = ctx.heads
        nqk = ctx.sizes[0] + ctx.sizes[1]
        gw = torch.empty(2 * d, dtype=y.dtype, device=dev)
        # The three gradients as one matrix already (this package's attention kernels write them that way): the
        # kernel then works in place on the query and key blocks and the value block is where it belongs.
        # In place only on gradients nobody else was handed: the caller (`qkv_norm_rope`) has named the outputs,
        # and none of them carries a tensor hook, whose function receives these very tensors before this one runs
        # and may have kept them.  (`retain_grad` keeps a copy of its own.)
        refs = getattr(ctx, "prl_outputs", None)
        watched = refs is None or any(z is not None and z._backward_hooks for z in (r() for r in refs))
        gy = None if watched else _as_one_matrix(gq, gk, gv, t, hq, hk, d, y)
        in_place = gy is not None
        if not in_place:
            gy = torch.empty_like(y)

        def rows(g: Tensor, h: int) -> Tensor:  # [1, h, T, d] -> [T, h, d] with 16-byte aligned strides
            g = g.transpose(1, 2).reshape(t, h, d)
            es = g.element_size()
            if g.stride(2) != 1 or (g.stride(0) * es) % 16 or (g.stride(1) * es) % 16 or g.data_ptr() % 16:
                g = g.contiguous()
            return g

        if in_place:
            gq3, gk3 = gy[:, : hq * d].view(t, hq, d), gy[:, hq * d : nqk].view(t, hk, d)
        else:
            gq3 = rows(gq, hq) if gq is not None else torch.zeros((t, hq, d), dtype=y.dtype, device=dev)
            gk3 = rows(gk, hk) if gk is not None else torch.zeros((t, hk, d), dtype=y.dtype, device=dev)
        code = C.dtype_code(y)
        ws_bytes = lib.prl_qk_norm_rope_bwd_workspace_bytes(code, t, hq, hk)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        s2 = lambda g: (C.c_int64 * 2)(g.stride(0), g.stride(1))
        with C.on_device(dev), C.timed("prl_qk_norm_rope_bwd", dev):
            C.check(lib.prl_qk_norm_rope_bwd(C.ptr(gq3), s2(gq3), C.ptr(gk3), s2(gk3), C.ptr(y), y.stride(0), C.ptr(nq),
                                             C.ptr(nk), C.ptr(rstd), C.ptr(cos2), C.ptr(sin2), cos2.stride(0),
                                             C.ptr(gy), gy.stride(0), C.ptr(gw), C.ptr(ws), ws_bytes, code, t, hq, hk,
                                             C.stream_ptr(dev)), "prl_qk_norm_rope_bwd")
        if in_place:
            pass
        elif gv is not None:
            gy[:, nqk:].copy_(gv.transpose(1, 2).reshape(t, ctx.sizes[2]))
        else:
            gy[:, nqk:].zero_()
        dx = mm_nn(gy, stacked).view(*ctx.lead, x2.shape[1]) if ctx.needs_input_grad[0] else None
        dw = weight_grad(gy, x2).split(ctx.sizes, dim=0)
        return dx, None, dw[0], dw[1], dw[2], gw[:d], gw[d:], None, None, None


def qkv_norm_rope(x: Tensor, stacked: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, nq: Tensor, nk: Tensor, cos: Tensor,
                  sin: Tensor, eps: float) -> tuple[Tensor, Tensor, Tensor]:
    """`_QKVNormRope.apply`, with the three outputs made known to the backward node: it works in place on their
    gradients (when those arrive as the blocks of one matrix) only if it can see that no output carries a hook."""
    import weakref

    q, k, v = _QKVNormRope.apply(x, stacked, wq, wk, wv, nq, nk, cos, sin, eps)
    node = q.grad_fn
    if node is not None and node is k.grad_fn and node is v.grad_fn:
        node.prl_outputs = tuple(weakref.ref(z) for z in (q, k, v))
    return q, k, v


def _fused_attention_forward(self: nn.Module, hidden_states: Tensor, position_embeddings: Any,
                             attention_mask: Tensor | None, past_key_values: Any = None, **kwargs: Any):
    """The forward of a transformers Qwen3-style attention block with the steps before the attention call taken
    by `_QKVNormRope` when they qualify (one packed row, stacked q/k/v weights, head dimension 128, per-head
    query / key norms); anything else goes to the block's own forward."""
    x = hidden_states
    cos, sin = position_embeddings
    qn, kn = getattr(self, "q_norm", None), getattr(self, "k_norm", None)
    ok = past_key_values is None and x.ndim == 3 and x.shape[0] == 1 and self.head_dim == 128 \
        and qn is not None and kn is not None and hasattr(qn, "variance_epsilon") \
        and qn.variance_epsilon == kn.variance_epsilon and not hasattr(qn.weight, "to_local") \
        and cos.dtype == x.dtype and cos.shape == (1, x.shape[1], 128) and cos.stride(-1) == 1 \
        and sin.stride() == cos.stride() and not cos.requires_grad \
        and (cos.stride(1) * cos.element_size()) % 16 == 0 and cos.data_ptr() % 16 == 0 and sin.data_ptr() % 16 == 0
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS

    stacked = _stackable(x, self.q_proj, self.k_proj, self.v_proj) if ok else None
    if stacked is None or self.config._attn_implementation not in ALL_ATTENTION_FUNCTIONS \
            or not C.load().prl_qk_norm_rope_supported(C.dtype_code(x), self.head_dim):
        return self._prl_torch_forward(hidden_states, position_embeddings, attention_mask, past_key_values, **kwargs)
    q, k, v = qkv_norm_rope(x, stacked, self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, qn.weight,
                            kn.weight, cos, sin, float(qn.variance_epsilon))
    attend = ALL_ATTENTION_FUNCTIONS[self.config._attn_implementation]  # what the block's own forward looks up
    out, weights = attend(self, q, k, v, attention_mask, dropout=0.0 if not self.training else self.attention_dropout,
                          scaling=self.scaling, sliding_window=getattr(self, "sliding_window", None), **kwargs)
    out = out.reshape(*x.shape[:-1], -1).contiguous()
    return self.o_proj(out), weights


class _QKVShare:
    """What the three projections of one attention module share: the query projection, called first, runs the
    stacked product and leaves the key and value pieces for the two calls that follow on the same input."""

    def __init__(self, attn: nn.Module):
        self.attn = attn
        self.x: Tensor | None = None
        self.k: Tensor | None = None
        self.v: Tensor | None = None


def _q_forward(self: nn.Module, x: Tensor) -> Tensor:
    share: _QKVShare = self._prl_qkv
    attn = share.attn
    stacked = _stackable(x, attn.q_proj, attn.k_proj, attn.v_proj)
    if stacked is None:
        share.x = None
        return self._prl_torch_forward(x)
    q, share.k, share.v = _LinearSplit.apply(x, stacked, attn.q_proj.weight, attn.k_proj.weight, attn.v_proj.weight)
    share.x = x
    return q


def _k_forward(self: nn.Module, x: Tensor) -> Tensor:
    share: _QKVShare = self._prl_qkv
    if share.x is x and share.k is not None:
        k, share.k = share.k, None
        return k
    return self._prl_torch_forward(x)


def _v_forward(self: nn.Module, x: Tensor) -> Tensor:
    share: _QKVShare = self._prl_qkv
    if share.x is x and share.v is not None:
        v, share.v, share.x = share.v, None, None
        return v
    share.x = None
    return self._prl_torch_forward(x)


def _lm_head_forward(self: nn.Module, hidden_states: Tensor) -> Tensor:
    from . import lm_head as LM

    return LM.linear(hidden_states, self.weight)


def fuse_model(model: nn.Module) -> dict[str, int]:
    """Rebind the forward of every RMSNorm module of an HF causal LM to the HIP kernel.  Returns what was
    rebound, by kind.  Call it before FSDP wrapping or after; parameters are left as they are."""
    done = {"rmsnorm": 0, "rope": 0, "swiglu": 0, "qkv": 0, "linear": 0, "layer": 0, "lm_head": 0}
    import os
    import sys

    layers_too = os.environ.get("PRL_FUSED_LAYER", "1") != "0"  # the residual additions inside the norm kernels

    home = sys.modules.get(type(model).__module__)  # e.g. transformers.models.qwen3.modeling_qwen3
    if home is not None and hasattr(home, "apply_rotary_pos_emb"):
        if home.__name__ not in _PATCHED_MODULES:
            _PATCHED_MODULES[home.__name__] = home.apply_rotary_pos_emb
            if not _ORIGINAL_ROPE:
                _ORIGINAL_ROPE.append(home.apply_rotary_pos_emb)
            home.apply_rotary_pos_emb = apply_rotary_pos_emb  # module-wide: every model of this family in the process
        done["rope"] = 1
    head = getattr(model, "lm_head", None)  # its products on this package's MFMA kernels where they are ahead
    if isinstance(head, nn.Linear) and head.bias is None and not hasattr(head, "_prl_torch_forward"):
        head._prl_torch_forward = head.forward
        head.forward = types.MethodType(_lm_head_forward, head)
        done["lm_head"] = 1
    for module in model.modules():
        name = type(module).__name__
        is_norm = name.endswith("RMSNorm") and hasattr(module, "weight") and hasattr(module, "variance_epsilon")
        if is_norm and not hasattr(module, "_prl_torch_forward"):
            module._prl_torch_forward = module.forward
            module.forward = types.MethodType(_fused_rmsnorm_forward, module)
            done["rmsnorm"] += 1
        if layers_too and _is_plain_decoder_layer(module) and not hasattr(module, "_prl_torch_forward"):
            module._prl_torch_forward = module.forward
            module._prl_carry, module._prl_first = None, False
            module.forward = types.MethodType(_fused_layer_forward, module)
            done["layer"] += 1
        is_gated_mlp = name.endswith("MLP") and all(hasattr(module, a) for a in ("gate_proj", "up_proj", "down_proj")) \
            and type(getattr(module, "act_fn", None)).__name__ in ("SiLU", "SiLUActivation")
        if is_gated_mlp and not hasattr(module, "_prl_torch_forward"):
            module._prl_torch_forward = module.forward
            module.forward = types.MethodType(_fused_mlp_forward, module)
            done["swiglu"] += 1
        # The blocks' other linear layers, one by one: o and down always; gate and up for the passes in which they are
        # not one stacked product (`_fused_mlp_forward` calls the modules then); q, k and v where they carry a bias
        # (no stacked product for those).  Each product goes where `lm_head.layer_plan` has it measured.
        one_by_one = ["o_proj", "down_proj"] + (["gate_proj", "up_proj"] if is_gated_mlp else [])
        if name.endswith("Attention") and any(getattr(getattr(module, a, None), "bias", None) is not None
                                              for a in ("q_proj", "k_proj", "v_proj")):
            one_by_one += ["q_proj", "k_proj", "v_proj"]
        for attr in one_by_one:
            lin = getattr(module, attr, None)
            if isinstance(lin, nn.Linear) and not hasattr(lin, "_prl_torch_forward") \
                    and (name.endswith("Attention") or name.endswith("MLP")):
                lin._prl_torch_forward = lin.forward
                lin.forward = types.MethodType(_split_dw_linear_forward, lin)
                done["linear"] += 1
        projections = [getattr(module, a, None) for a in ("q_proj", "k_proj", "v_proj")]
        if name.endswith("Attention") and all(isinstance(m, nn.Linear) and m.bias is None for m in projections) \
                and not hasattr(projections[0], "_prl_torch_forward"):
            share = _QKVShare(module)
            for m, fn in zip(projections, (_q_forward, _k_forward, _v_forward)):
                m._prl_torch_forward = m.forward
                m._prl_qkv = share
                m.forward = types.MethodType(fn, m)
            done["qkv"] += 1
            if hasattr(module, "q_norm") and hasattr(module, "k_norm") and hasattr(module, "head_dim") \
                    and hasattr(module, "o_proj") and not hasattr(module, "_prl_torch_forward"):
                module._prl_torch_forward = module.forward
                module.forward = types.MethodType(_fused_attention_forward, module)
    return done


def unfuse_model(model: nn.Module) -> None:
    import sys

    home = sys.modules.get(type(model).__module__)
    if home is not None and home.__name__ in _PATCHED_MODULES:
        home.apply_rotary_pos_emb = _PATCHED_MODULES.pop(home.__name__)
    for module in model.modules():
        if hasattr(module, "_prl_torch_forward"):
            module.forward = module._prl_torch_forward
            del module._prl_torch_forward
            for attr in ("_prl_qkv", "_prl_carry", "_prl_first"):
                if hasattr(module, attr):
                    delattr(module, attr)


def describe(model: nn.Module) -> Any:
    return sum(1 for m in model.modules() if hasattr(m, "_prl_torch_forward"))
