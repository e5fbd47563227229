"""Cost of gathering the hidden states of the scored positions in front of the LM head and of scattering their
gradient back (what `logits_to_keep=positions` adds to a step), against the LM-head work it removes."""
import torch, time
dev = torch.device("cuda:0")
for T, H, frac in ((16384, 1024, 0.86), (16384, 1024, 0.5), (4096, 3584, 0.66), (16384, 3584, 0.66)):
    h = torch.randn(1, T, H, device=dev, dtype=torch.bfloat16, requires_grad=True)
    idx = torch.randperm(T - 1, device=dev)[: int(frac * T) // 256 * 256].sort().values
    g = torch.randn(1, idx.numel(), H, device=dev, dtype=torch.bfloat16)
    def run():
        h.grad = None
        y = h[:, idx, :]
        y.backward(g)
    def run2():
        h.grad = None
        y = h.index_select(1, idx)
        y.backward(g)
    for name, f in (("h[:, idx, :]", run), ("index_select", run2)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            f()
        b.record()
        torch.cuda.synchronize()
        print(f"T={T} H={H} rows={idx.numel()} {name}: {a.elapsed_time(b) / 10 * 1e3:.0f} us forward + backward", flush=True)
