"""A transformer block's linear layers (forward, input gradient, weight gradient) over T rows, for a few T around a
joined row's length: whether the GEMM library has sizes it serves markedly better.  Qwen3-0.6B dimensions by default.

    python3 tools/block_rows_sweep.py [hidden] [intermediate] [q_out] [kv_out]
"""
import sys

import torch

arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
H, I, Q, KV = arg(1, 1024), arg(2, 3072), arg(3, 2048), arg(4, 1024)
dev = torch.device("cuda:0")
layers = {"q": (H, Q), "k": (H, KV), "v": (H, KV), "o": (Q, H), "gate": (H, I), "up": (H, I), "down": (I, H)}
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=8):
    fn()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for t in (8192, 10240, 12288, 13312, 14336, 14848, 15360, 15872, 16128, 16384):
    total, parts = 0.0, []
    for name, (k, n) in layers.items():
        w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
        x = torch.randn(t, k, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(t, n, device=dev, dtype=torch.bfloat16)
        f, dx, dw = timed(lambda: torch.nn.functional.linear(x, w)), timed(lambda: dy @ w), timed(lambda: dy.t() @ x)
        total += f + dx + dw
        parts.append(f"{name} {1e3 * f:4.0f}/{1e3 * dx:4.0f}/{1e3 * dw:4.0f}")
    flop = 6.0 * t * sum(k * n for k, n in layers.values())
    print(f"T {t:6d}: {total:6.3f} ms = {1e3 * total / t * 1024:6.1f} us per 1024 rows, {flop / total / 1e9:5.0f} TF/s | "
          + " ".join(parts), flush=True)
