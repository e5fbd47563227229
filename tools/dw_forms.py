"""The weight gradient dW[V, H] = dY[T, V]^T X[T, H] in its two forms: both operands read along the tokens by
transposing LDS reads (the default where the weight is not very wide), and a transposed copy of the narrower operand
in memory first (PRL_LM_HEAD_DW_FORM=transpose; the default for the LM head itself), in turn, medians of 5:
    python3 tools/dw_forms.py [T]"""
import os
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import lm_head as L  # noqa: E402

t = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
SHAPES = [(1024, 4096), (2048, 1024), (1024, 6144), (3072, 1024), (2048, 12288), (6144, 2048), (3584, 4608), (3584, 37888),
          (18944, 3584), (5120, 55296), (1024, 16384), (1024, 32768), (1024, 151936), (2048, 151936), (3584, 152064),
          (5120, 152064)]
dev = torch.device("cuda:0")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=3):
    fn()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for h, v in SHAPES:
    x = torch.randn(t, h, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(t, v, device=dev, dtype=torch.bfloat16)
    times = {"both": [], "transpose": []}
    for _ in range(5):
        for form in ("both", "transpose"):
            os.environ["PRL_LM_HEAD_DW_FORM"] = form
            times[form].append(timed(lambda: L.lm_head_bwd_weight(dy, x)))
    m = {k: statistics.median(v_) for k, v_ in times.items()}
    print(f"{h:5d} -> {v:6d}  V/H {v / h:6.1f}  read along the tokens {m['both']:.3f} ms  transposed copy {m['transpose']:.3f} ms  "
          f"ratio {m['transpose'] / m['both']:.3f}", flush=True)
    del x, dy
