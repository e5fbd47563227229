"""PyTorch-ROCm's two flash-attention libraries (AOTriton, the default, and composable_kernel, if this build has it)
on a wordle-shaped row: 4 samples of 4096 tokens, 16 query / 8 key-value heads of 128, bf16, causal; through
`varlen_attn` on the packed row and through SDPA on the samples side by side.  Forward + backward, HIP events.

    python3 tools/fa_backend_probe.py
"""
import sys, statistics
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel
from torch.nn.attention.varlen import varlen_attn

dev = torch.device("cuda:0")
n, L, hq, hk, d = 4, 4096, 16, 8, 128
t = n * L
gen = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(t, hq, d, generator=gen, device=dev).to(torch.bfloat16).requires_grad_(True)
k = torch.randn(t, hk, d, generator=gen, device=dev).to(torch.bfloat16).requires_grad_(True)
v = torch.randn(t, hk, d, generator=gen, device=dev).to(torch.bfloat16).requires_grad_(True)
dout = torch.randn(t, hq, d, generator=gen, device=dev).to(torch.bfloat16)
cu = torch.arange(0, t + 1, L, dtype=torch.int32, device=dev)
q4, k4, v4 = (x.detach().view(n, L, -1, d).transpose(1, 2).requires_grad_(True) for x in (q, k, v))
d4 = dout.view(n, L, hq, d).transpose(1, 2)


def varlen():
    o = varlen_attn(q, k, v, cu, cu, L, L, is_causal=True)
    o.backward(dout)
    return o


def sdpa():
    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
        o = F.scaled_dot_product_attention(q4, k4, v4, is_causal=True, enable_gqa=True)
    o.backward(d4)
    return o


def timed(fn):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)


from prime_rl_cse_amd import attention

pos = torch.arange(L).repeat(n)[None]
plan = attention.plan_from_position_ids(pos, dev)
qh, kh, vh = (x.detach().transpose(0, 1)[None].requires_grad_(True) for x in (q, k, v))  # [1, H, T, d] views


def in_tree():
    o = attention.long_attention(qh, kh, vh, plan)
    o.backward(dout[None])
    return o


def in_tree_forward():
    with torch.no_grad():
        return attention.long_attention(qh, kh, vh, plan)


def varlen_forward():
    with torch.no_grad():
        return varlen_attn(q, k, v, cu, cu, L, L, is_causal=True)


flop = 3.5 * n * hq * (2 * 2 * L * L * d) / 2  # forward + backward (2.5 x), causal half
ref = None
o = in_tree().detach().float().reshape(-1)[:4096 * 128].clone()
ref = varlen().detach().float().reshape(-1)[:4096 * 128].clone()
us = timed(in_tree)
print(f"this package's tiled kernels           {us:8.0f} us forward + backward = {flop / us / 1e6:6.1f} TFLOP/s; max |difference| to "
      f"varlen_attn {float((o - ref).abs().max()):.2e}", flush=True)
us_f, us_vf = timed(in_tree_forward), timed(varlen_forward)
print(f"forward alone: this package {us_f:.0f} us = {flop / 3.5 / us_f / 1e6:.1f} TFLOP/s, varlen_attn {us_vf:.0f} us = "
      f"{flop / 3.5 / us_vf / 1e6:.1f} TFLOP/s", flush=True)
for lib in ("aotriton", "ck"):
    try:
        torch.backends.cuda.preferred_rocm_fa_library(lib)
        print("library:", torch.backends.cuda.preferred_rocm_fa_library(), flush=True)
    except Exception as e:
        print(f"library {lib}: not available in this build ({e!r})"[:300], flush=True)
        continue
    for name, fn in (("varlen_attn on the packed row", varlen), ("SDPA flash on [4, H, 4096, 128]", sdpa)):
        try:
            o = fn().detach().float().reshape(-1)[:4096 * 128].clone()
            us = timed(fn)
            if ref is None:
                ref = o
            print(f"  {name:34s} {us:8.0f} us forward + backward = {flop / us / 1e6:6.1f} TFLOP/s; max |difference| to the "
                  f"first run {float((o - ref[: o.numel()]).abs().max()) if name.startswith('varlen') else float('nan'):.2e}",
                  flush=True)
        except Exception as e:
            print(f"  {name}: failed ({e!r})"[:400], flush=True)
