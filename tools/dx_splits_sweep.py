"""The LM head's input gradient over R rows with the K split count forced to 1 .. 8 (PRL_LM_HEAD_DX_SPLITS), next to
the count the library's rule picks:   python3 tools/dx_splits_sweep.py R [H] [V]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import lm_head as L  # noqa: E402

r = int(sys.argv[1])
h, v = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 1024), (3, 151936)))
dev = torch.device("cuda:0")
w = torch.randn(v, h, device=dev, dtype=torch.bfloat16) * 0.02
dy = torch.randn(r, v, device=dev, dtype=torch.bfloat16)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(reps=5):
    L.lm_head_bwd_input(dy, w)
    a.record()
    for _ in range(reps):
        L.lm_head_bwd_input(dy, w)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


os.environ.pop("PRL_LM_HEAD_DX_SPLITS", None)
print(f"R {r} H {h}: the rule's own choice {timed():.3f} ms", flush=True)
for s in range(1, 9):
    os.environ["PRL_LM_HEAD_DX_SPLITS"] = str(s)
    print(f"  splits {s}: {timed():.3f} ms", flush=True)
