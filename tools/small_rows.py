"""Single-pass against two-pass for short packed rows (fewer rows than compute units): time of policy_loss + backward."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import prime_rl_cse_amd as prl

dev = "cuda:0"
v = 151936
for t in (16, 64, 128, 192, 256, 320, 384, 512, 768, 1024):
    g = torch.Generator(device=dev).manual_seed(t)
    logits = (torch.randn(1, t, v, generator=g, device=dev) * 2.5).to(torch.bfloat16)
    ids = torch.randint(0, v, (1, t), generator=g, device=dev)
    pos = torch.arange(t, device=dev)[None]
    adv = torch.randn(1, t, generator=g, device=dev)
    old = torch.full((1, t), -12.0, device=dev)
    mask = torch.ones(1, t, dtype=torch.bool, device=dev)
    res = {}
    for flag in ("1", "0"):
        os.environ["PRL_FUSED_ROWS"] = flag
        def once():
            x = logits.detach().requires_grad_(True)
            loss, _ = prl.policy_loss(x, ids, pos, adv, old, mask, 1.0, prl.LossConfig(), t)
            loss.backward(prl.unit_gradient(dev))
        for _ in range(5): once()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): once()
        b.record(); torch.cuda.synchronize()
        res[flag] = a.elapsed_time(b) / 30 * 1e3
    print(f"T={t:5d}: single pass {res['1']:7.1f} us   two passes {res['0']:7.1f} us")
os.environ.pop("PRL_FUSED_ROWS", None)
