"""Host time of the pieces of one micro-batch of the headline loop, on a row so short (128 tokens) that the GPU
is never what the host waits for: `policy_loss` + backward, the seven `Tensors.add` calls, `compute_stats_async`
over 8 micro-batches' pieces; then cProfile of the last.

    python3 tools/loop_host_cost.py
"""
import cProfile, pstats, sys, time
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
unit = prl.unit_gradient(dev)


def loss_bwd():
    x = logits.requires_grad_(True)
    x.grad = None
    loss, aux = prl.policy_loss(x, ids, pos, adv, old, mask, 1.0, cfg, t)
    loss.backward(unit)
    return loss, aux


def adds(ts, loss, aux):
    ts.add("probs", aux["logprobs"], mask, exp=True)
    ts.add("old_probs", old, mask, exp=True)
    ts.add("entropy", aux["entropy"], mask)
    ts.add("loss", loss.detach().reshape(1))
    for key in ("importance_ratio", "clipped_importance_ratio", "is_clipped"):
        ts.add(key, aux[key], mask)


def step(clock):
    ts = prl.Tensors()
    for _ in range(8):
        h0 = time.perf_counter()
        loss, aux = loss_bwd()
        h1 = time.perf_counter()
        adds(ts, loss, aux)
        h2 = time.perf_counter()
        clock[0] += h1 - h0
        clock[1] += h2 - h1
    h0 = time.perf_counter()
    fut = ts.compute_stats_async()
    clock[2] += time.perf_counter() - h0
    return fut


for _ in range(10):
    step([0, 0, 0]).result()
torch.cuda.synchronize()
clock = [0.0, 0.0, 0.0]
n = 50
prev = None
for _ in range(n):
    fut = step(clock)
    if prev is not None:
        prev.result()
    prev = fut
torch.cuda.synchronize()
print(f"policy_loss + backward {clock[0] / n / 8 * 1e6:6.1f} us per micro-batch; seven Tensors.add {clock[1] / n / 8 * 1e6:6.1f} us; "
      f"compute_stats_async {clock[2] / n * 1e6:6.1f} us per step", flush=True)
ts_list = []
for _ in range(40):
    ts = prl.Tensors()
    for _ in range(8):
        loss, aux = loss_bwd()
        adds(ts, loss, aux)
    ts_list.append(ts)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
futs = [ts.compute_stats_async() for ts in ts_list]
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    loss_bwd()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
