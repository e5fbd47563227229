"""Time of the short-sample attention kernels against PyTorch-ROCm's SDPA on the side-by-side layout and its varlen_attn, forward and
backward, for rows of 128-token samples (Qwen3-0.6B heads: 16 query, 8 key / value, dimension 128).

    python tools/attn_bench.py [samples] [reps]        # also a convenient target for rocprofv3 --pmc
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.nn.functional as F

from prime_rl_cse_amd import attention

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L, hq, hk, d = 128, 16, 8, 128
t = n * L
gen = torch.Generator(device=dev).manual_seed(0)
mk = lambda h: torch.randn(1, t, h, d, generator=gen, device=dev).to(torch.bfloat16).transpose(1, 2).requires_grad_(True)
q, k, v = mk(hq), mk(hk), mk(hk)
dout = torch.randn(1, t, hq, d, generator=gen, device=dev).to(torch.bfloat16)
pos = torch.arange(L).repeat(n)[None]
plan = attention.plan_from_position_ids(pos, dev)


def ours():
    out = attention.short_attention(q, k, v, plan, d ** -0.5)
    out.backward(dout)


def sdpa():
    side = lambda x: x.squeeze(0).unflatten(1, (n, L)).permute(1, 0, 2, 3)
    out = F.scaled_dot_product_attention(side(q), side(k), side(v), is_causal=True, enable_gqa=True)
    out.permute(0, 2, 1, 3).reshape(1, t, hq, d).backward(dout)


def varlen():
    out = attention.varlen_attention(q, k, v, plan)
    out.backward(dout)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print(f"{n} samples of {L} tokens: short-sample kernels {timed(ours):.1f} us, SDPA side by side {timed(sdpa):.1f} us, "
      f"torch varlen_attn {timed(varlen):.1f} us (forward + backward, host launch overhead included)")
