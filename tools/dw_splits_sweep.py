"""The weight gradient dW[V, H] = dY[T, V]^T X[T, H] on the MFMA kernels with the K split count forced to 1 .. 16
(PRL_LM_HEAD_DW_SPLITS), next to the count the library's rule picks and to `fused.weight_grad` (the BLAS library over
ranges of the tokens), on the shapes of the decoder blocks' linear layers:
    python3 tools/dw_splits_sweep.py [T]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import fused, lm_head as L  # noqa: E402

os.environ["PRL_LAYER_GEMMS"] = "lib"  # `fused.weight_grad` below is the library's side of the comparison
t = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
SHAPES = [(1024, 4096), (2048, 1024), (1024, 6144), (3072, 1024), (2048, 4096), (2048, 2048), (2048, 12288), (6144, 2048),
          (3584, 4608), (3584, 3584), (5120, 5120)]
dev = torch.device("cuda:0")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=5):
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
    os.environ.pop("PRL_LM_HEAD_DW_SPLITS", None)
    line = [f"{h:5d} -> {v:5d}: library {timed(lambda: fused.weight_grad(dy, x)):.3f}  rule {timed(lambda: L.lm_head_bwd_weight(dy, x)):.3f} |"]
    for s in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
        os.environ["PRL_LM_HEAD_DW_SPLITS"] = str(s)
        line.append(f"{s}: {timed(lambda: L.lm_head_bwd_weight(dy, x)):.3f}")
    print("  ".join(line), flush=True)
