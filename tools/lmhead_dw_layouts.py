"""The LM head's weight gradient dW [V, H] = dlogits^T [V, T] x hidden [T, H] at the joined row's size
(T = 16384, V = 151936, H = 1024, bf16) in the layouts it can be asked for, HIP events, random order.

    python3 tools/lmhead_dw_layouts.py
"""
import random, statistics, sys
import torch

dev = torch.device("cuda:0")
T, V, H = 16384, 151936, 1024
g = torch.Generator(device=dev).manual_seed(0)
gy = torch.empty(T, V, dtype=torch.bfloat16, device=dev)
for r in range(0, T, 1024):
    gy[r : r + 1024] = (torch.randn(1024, V, generator=g, device=dev) * 1e-3).to(torch.bfloat16)
x = torch.randn(T, H, generator=g, device=dev).to(torch.bfloat16)
w = torch.randn(V, H, generator=g, device=dev).to(torch.bfloat16)


def a():  # as autograd asks for it: both operands summed over their row index
    return gy.t() @ x


def b():  # hidden transposed first (33 MB), result [H, V], then transposed back (311 MB)
    return (x.t().contiguous() @ gy).t().contiguous()


def c():  # the same without making the result contiguous
    return x.t().contiguous() @ gy


def d():  # split over the tokens into 2 batched products and their sum
    s = 2
    return torch.bmm(gy.view(s, T // s, V).transpose(1, 2), x.view(s, T // s, H)).sum(0)


def fwd():
    return x @ w.t()


def dx():
    return gy @ w


fns = {"gy.t() @ x": a, "(x.t().contiguous() @ gy).t().contiguous()": b, "x.t().contiguous() @ gy  [H, V]": c,
       "2 batched products over the tokens + sum": d, "forward x @ w.t()": fwd, "dx = gy @ w": dx}
ref = a().float()
for name, fn in fns.items():
    if name.startswith(("forward", "dx")):
        continue
    out = fn()
    out = out.t() if out.shape == (H, V) else out
    err = (out.float() - ref).abs().max().item()
    print(f"{name:46s} max |difference to the first| {err:.3e}", flush=True)
    del out
del ref
times = {k: [] for k in fns}
order = list(fns)
for _ in range(6):
    random.shuffle(order)
    for k in order:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fns[k]()
        e0.record()
        for _ in range(2):
            fns[k]()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 2)
flop = 2.0 * T * V * H
for k in fns:
    ms = statistics.median(times[k])
    print(f"{k:46s} {ms:6.2f} ms  {flop / ms / 1e12:5.2f} PFLOP/s", flush=True)
