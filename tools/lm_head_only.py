"""Launches only the LM head's products, for profilers:  python tools/lm_head_only.py [reps] [T] [H] [V] [who]

`who`: kernel (this package's MFMA kernels), library (torch's products), both (default).  Random bf16 operands."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import lm_head as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t, h, v = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (16384, 1024, 151936)
who = sys.argv[5] if len(sys.argv) > 5 else "both"
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((t, h), generator=gen, device=dev).to(torch.bfloat16)
w = (torch.randn((v, h), generator=gen, device=dev) * 0.05).to(torch.bfloat16)
g = torch.empty((t, v), dtype=torch.bfloat16, device=dev)
for r in range(0, t, 2048):
    g[r:r + 2048] = (torch.randn((min(2048, t - r), v), generator=gen, device=dev) * 0.01).to(torch.bfloat16)
torch.cuda.synchronize()
for _ in range(reps):
    if who in ("kernel", "both"):
        L.lm_head_fwd(x, w)
        L.lm_head_bwd_input(g, w)
        L.lm_head_bwd_weight(g, x)
    if who in ("library", "both"):
        torch.nn.functional.linear(x, w)
        g @ w
        g.t() @ x
torch.cuda.synchronize()
print("done", reps, t, h, v, who)
