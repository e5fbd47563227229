"""GPU time of the transformer blocks' GEMMs at T = 2048, H = 1024 (Qwen3-0.6B): the separate q / k / v and
gate / up products against one product over their concatenated weights (forward, dX and dW shapes)."""
import torch

dev = torch.device("cuda:0")
T, H = 2048, 1024


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


x = torch.randn(T, H, device=dev, dtype=torch.bfloat16)
for name, parts in (("q,k,v", (2048, 1024, 1024)), ("gate,up", (3072, 3072))):
    ws = [torch.randn(n, H, device=dev, dtype=torch.bfloat16) for n in parts]
    wf = torch.cat(ws)
    gys = [torch.randn(T, n, device=dev, dtype=torch.bfloat16) for n in parts]
    gyf = torch.cat(gys, dim=1)
    fwd_sep = sum(timed(lambda w=w: x @ w.t()) for w in ws)
    fwd_one = timed(lambda: x @ wf.t())
    dx_sep = sum(timed(lambda g=g, w=w: g @ w) for g, w in zip(gys, ws))
    dx_one = timed(lambda: gyf @ wf)
    dw_sep = sum(timed(lambda g=g: g.t() @ x) for g in gys)
    dw_one = timed(lambda: gyf.t() @ x)
    add = timed(lambda: x + x)
    cat = timed(lambda: torch.cat(gys, dim=1))
    print(f"{name:8s} forward {fwd_sep:6.1f} -> {fwd_one:6.1f} us | dX {dx_sep:6.1f} (+{(len(parts) - 1) * add:4.1f} adds) -> "
          f"{dx_one:6.1f} us | dW {dw_sep:6.1f} -> {dw_one:6.1f} us | cat of the gradients {cat:5.1f} us")
