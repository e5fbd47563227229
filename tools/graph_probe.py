"""Can one micro-batch of the trainer (HF body + HIP epilogue, forward and backward) be captured in a HIP graph?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import prime_rl_cse_amd as prl
from prime_rl_cse_amd import trainer as T

dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=layers)
model = T.get_model(cfg, dev, torch.bfloat16, seed=0)
t = 2048
g = torch.Generator(device=dev).manual_seed(0)
ids = torch.randint(0, 100, (1, t), generator=g, device=dev)
pos = torch.arange(128, device=dev).repeat(16).unsqueeze(0)
adv = torch.randn(1, t, generator=g, device=dev)
old = torch.randn(1, t, generator=g, device=dev)
mask = torch.ones(1, t, dtype=torch.bool, device=dev)
loss_cfg = prl.LossConfig()

def one():
    logits = T.forward(model, ids, pos)
    loss, aux = prl.policy_loss(logits, ids, pos, adv, old, mask, 1.0, loss_cfg, 16384, skip_masked=True)
    loss.backward()
    return loss, aux

# eager timing
for _ in range(3): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): one()
torch.cuda.synchronize(); print(f"eager {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per micro-batch ({layers} layers)")
ref = [p.grad.clone() for p in model.parameters()]

# capture
for p in model.parameters(): p.grad.zero_()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): one()
torch.cuda.current_stream().wait_stream(s)
for p in model.parameters(): p.grad.zero_()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss, aux = one()
torch.cuda.synchronize()
for p in model.parameters(): p.grad.zero_()
for _ in range(8): graph.replay()
torch.cuda.synchronize()
got = [p.grad.clone() for p in model.parameters()]
err = max(((a - b).abs().max() / (b.abs().max() + 1e-12)).item() for a, b in zip(got, ref))
print("grad agreement after 8 replays vs 8 eager accumulations: max rel diff", err)
t0 = time.perf_counter()
for _ in range(5): graph.replay()
torch.cuda.synchronize(); print(f"graph {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per micro-batch; loss {loss.item():.6f}")
