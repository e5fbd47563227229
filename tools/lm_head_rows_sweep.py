"""The LM head's three products (logits, input gradient, weight gradient) over R gathered rows, R in steps of 256:
which row counts the GEMM library serves at full rate, and what this package's kernels take for the same rows.
H = 1024, V = 151936 (Qwen3-0.6B), bf16.

    python3 tools/lm_head_rows_sweep.py [lo] [hi] [step] [H] [V]
"""
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import lm_head as L  # noqa: E402

lo, hi, step = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8192), (2, 16384), (3, 256)))
dev = torch.device("cuda:0")
H, V = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((4, 1024), (5, 151936)))
w = torch.randn(V, H, device=dev, dtype=torch.bfloat16) * 0.02
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=5):
    fn()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for r in range(lo, hi + 1, step):
    h = torch.randn(r, H, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(r, V, device=dev, dtype=torch.bfloat16)
    t_f = timed(lambda: torch.nn.functional.linear(h, w))
    t_x = timed(lambda: dy @ w)
    t_w = timed(lambda: dy.t() @ h)
    k_f = timed(lambda: L.lm_head_fwd(h, w))
    k_x = timed(lambda: L.lm_head_bwd_input(dy, w))
    k_w = timed(lambda: L.lm_head_bwd_weight(dy, h))
    flop = 2.0 * r * H * V
    print(f"R {r:6d}: logits {t_f:6.3f} ms {flop / t_f / 1e9:6.0f} TF/s | dX {t_x:6.3f} ms {flop / t_x / 1e9:6.0f} | "
          f"dW {t_w:6.3f} ms {flop / t_w / 1e9:6.0f} | all three {t_f + t_x + t_w:6.3f} ms, per 1024 rows "
          f"{(t_f + t_x + t_w) * 1024 / r:6.3f} || kernels: logits {k_f:6.3f} dX {k_x:6.3f} dW {k_w:6.3f} all three "
          f"{k_f + k_x + k_w:6.3f} ms, per 1024 rows {(k_f + k_x + k_w) * 1024 / r:6.3f}", flush=True)
    del h, dy
