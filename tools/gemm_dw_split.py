"""dW = gy^T @ x of the blocks' linear layers at 16384 tokens (a small output, K = 16384): the one product
autograd runs against the same sum split over the tokens into S batched products plus their sum, on operands that
rotate through more memory than the Infinity Cache holds."""
import torch

dev = torch.device("cuda:0")
M = 16384
shapes = {"qkv": (4096, 1024), "o": (1024, 2048), "gate_up": (6144, 1024), "down": (1024, 3072)}  # W [out, in]
SETS = 6


def timed(fn, n=30):
    for i in range(SETS):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i % SETS)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def split(gy, x, s, acc_dtype):
    p = torch.bmm(gy.view(s, M // s, -1).transpose(1, 2), x.view(s, M // s, -1), out_dtype=acc_dtype) \
        if acc_dtype is not None else torch.bmm(gy.view(s, M // s, -1).transpose(1, 2), x.view(s, M // s, -1))
    return p.sum(0, dtype=torch.float32).to(torch.bfloat16)


for name, (n_out, n_in) in shapes.items():
    gy = [torch.randn(M, n_out, device=dev, dtype=torch.bfloat16) for _ in range(SETS)]
    x = [torch.randn(M, n_in, device=dev, dtype=torch.bfloat16) for _ in range(SETS)]
    flop = 2 * M * n_out * n_in
    line = f"{name:8s} one product {timed(lambda i: gy[i].t() @ x[i]):6.1f} us"
    for s in (2, 4, 8):
        line += f" | S={s}: {timed(lambda i: split(gy[i], x[i], s, None)):6.1f}"
    try:
        for s in (4, 8):
            line += f" | S={s} fp32 partials: {timed(lambda i: split(gy[i], x[i], s, torch.float32)):6.1f}"
    except Exception as e:  # bmm(out_dtype=) is not in every build
        line += f" | fp32 partials: {type(e).__name__}"
    ref = (gy[0].float().t() @ x[0].float())
    err1 = ((gy[0].t() @ x[0]).float() - ref).norm() / ref.norm()
    err4 = (split(gy[0], x[0], 4, None).float() - ref).norm() / ref.norm()
    print(line + f" | rel. error one {err1:.2e}, S=4 {err4:.2e}")
    del gy, x
