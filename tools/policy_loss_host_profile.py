"""Where the host time of one policy_loss + backward goes (short row, so that the GPU is never the limit)."""
import cProfile, pstats, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import prime_rl_cse_amd as prl

dev = "cuda:0"
t, v = 128, 151936
logits = torch.randn(1, t, v, device=dev).to(torch.bfloat16)
ids = torch.randint(0, v, (1, t), device=dev)
pos = torch.arange(t, device=dev)[None]
adv = torch.randn(1, t, device=dev)
old = torch.full((1, t), -12.0, device=dev)
mask = torch.ones(1, t, dtype=torch.bool, device=dev)
cfg = prl.LossConfig()

def once():
    x = logits.detach().requires_grad_(True)
    loss, _ = prl.policy_loss(x, ids, pos, adv, old, mask, 1.0, cfg, t)
    loss.backward(prl.unit_gradient(dev))

for _ in range(20): once()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): once()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
